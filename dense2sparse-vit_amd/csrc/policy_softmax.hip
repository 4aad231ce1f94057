// Attention.softmax_with_policy (vit_models/dynamic_vit.py:195-214), forward and backward, on a materialised score
// tensor [B, H, N, N] with a keep policy [B, N]:
//     m_ij = policy[b,j] or (i == j);  e_ij = exp(a_ij - max_j a_ij) * m_ij;  out_ij = (e_ij + eps/N) / (sum_j e_ij + eps)
// One wave per score row, the row is held in registers (N <= 1024).  Backward includes the path through the row maximum
// (the eps terms make the function not exactly shift-invariant, and the reference does not detach the max).
#include "d2s_common.h"

namespace {

template <int NE>
__global__ __launch_bounds__(256) void policy_softmax_fwd_kernel(const float* __restrict__ attn, const float* __restrict__ policy,
                                                                 float* __restrict__ out, long rows, int H, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % N);
    const long b = row / ((long)H * N);
    const float* ar = attn + row * N;
    const float* pr = policy + b * N;
    float a[NE], m[NE];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int j = lane + u * 64;
        a[u] = j < N ? ar[j] : -INFINITY;
        m[u] = j < N ? (j == i ? 1.f : pr[j]) : 0.f;      // policy + (1 - policy) * eye
        mx = fmaxf(mx, a[u]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        a[u] = (lane + u * 64 < N) ? expf(a[u] - mx) * m[u] : 0.f;
        s += a[u];
    }
    s = wave_sum(s);
    const float inv = 1.0f / (s + eps), c = eps / (float)N;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int j = lane + u * 64;
        if (j < N) out[row * N + j] = (a[u] + c) * inv;
    }
}

template <int NE>
__global__ __launch_bounds__(256) void policy_softmax_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ policy,
                                                                 const float* __restrict__ gout, float* __restrict__ gattn, long rows,
                                                                 int H, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % N);
    const long b = row / ((long)H * N);
    const float* ar = attn + row * N;
    const float* pr = policy + b * N;
    float e[NE], g[NE], raw[NE];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int j = lane + u * 64;
        raw[u] = j < N ? ar[j] : -INFINITY;
        mx = fmaxf(mx, raw[u]);
    }
    mx = wave_max(mx);
    // first index attaining the maximum (torch.max(dim) returns one index; its backward routes the gradient there)
    int cand = 1 << 30;
#pragma unroll
    for (int u = 0; u < NE; ++u)
        if (raw[u] == mx) cand = min(cand, lane + u * 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int j = lane + u * 64;
        const float mk = j < N ? (j == i ? 1.f : pr[j]) : 0.f;
        e[u] = j < N ? expf(raw[u] - mx) * mk : 0.f;
        g[u] = j < N ? gout[row * N + j] : 0.f;
        s += e[u];
    }
    s = wave_sum(s);
    const float inv = 1.0f / (s + eps), c = eps / (float)N;
    // out_j = (e_j + c) * inv ;  d out_j / d e_k = [j==k] inv - (e_j + c) inv^2
    float gdot = 0.f, gsum_e = 0.f;
#pragma unroll
    for (int u = 0; u < NE; ++u) { gdot += g[u] * (e[u] + c) * inv; gsum_e += 0.f; }
    gdot = wave_sum(gdot);          // sum_j g_j out_j
    // d L / d e_k = inv * (g_k - gdot) ; d e_k / d a_k = e_k ; through the max: d e_k / d mx = -e_k for every k
    float through_max = 0.f;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const float de = inv * (g[u] - gdot);
        g[u] = de * e[u];           // direct path d L / d a_k
        through_max -= de * e[u];
    }
    through_max = wave_sum(through_max);
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int j = lane + u * 64;
        if (j < N) gattn[row * N + j] = g[u] + (j == cand ? through_max : 0.f);
    }
    (void)gsum_e;
}

}  // namespace

extern "C" {

// attn, out: [B,H,N,N]; policy: [B,N] (1 = kept token).  N <= 1024.
int d2s_softmax_policy_fwd(const float* attn, const float* policy, float* out, int B, int H, int N, float eps, hipStream_t stream) {
    if (!attn || !policy || !out || B <= 0 || H <= 0 || N <= 0 || N > 1024) return D2S_ERR_ARG;
    const long rows = (long)B * H * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int ne = (N + 63) / 64;
#define D2S_PS(NE) hipLaunchKernelGGL(policy_softmax_fwd_kernel<NE>, grid, block, 0, stream, attn, policy, out, rows, H, N, eps)
    if (ne <= 1) D2S_PS(1); else if (ne <= 2) D2S_PS(2); else if (ne <= 4) D2S_PS(4); else if (ne <= 8) D2S_PS(8); else D2S_PS(16);
#undef D2S_PS
    return d2s_check_launch();
}

int d2s_softmax_policy_bwd(const float* attn, const float* policy, const float* grad_out, float* grad_attn, int B, int H, int N, float eps,
                           hipStream_t stream) {
    if (!attn || !policy || !grad_out || !grad_attn || B <= 0 || H <= 0 || N <= 0 || N > 1024) return D2S_ERR_ARG;
    const long rows = (long)B * H * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int ne = (N + 63) / 64;
#define D2S_PS(NE) hipLaunchKernelGGL(policy_softmax_bwd_kernel<NE>, grid, block, 0, stream, attn, policy, grad_out, grad_attn, rows, H, N, eps)
    if (ne <= 1) D2S_PS(1); else if (ne <= 2) D2S_PS(2); else if (ne <= 4) D2S_PS(4); else if (ne <= 8) D2S_PS(8); else D2S_PS(16);
#undef D2S_PS
    return d2s_check_launch();
}

}  // extern "C"
