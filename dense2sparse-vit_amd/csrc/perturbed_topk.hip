// Perturbed (differentiable) top-k, vit_models/peturbed_topk.py:16-80, without the [b, nS, k, d] one-hot tensor
// (4.9 GB at b=128, nS=500, k=98, d=196 in the reference).
//
// forward : for every noise sample s, top-k of x + sigma * noise[s] (rank by counting in LDS), ids sorted ascending
//           by an ordered ballot compaction; the kk-th id j of a sample bumps an integer counter (kk, j).
//           indicators[b,kk,j] = count / nS.  Integer atomics -> exact and order independent.
// backward: grad_x[b,j] = 1/(nS*sigma) * sum_s noise[b,s,j] * [j selected in s] * g[b, pos_s(j), j]; the selection is
//           recomputed from (x, noise) instead of being stored; thread j owns grad_x[b,j], so no atomics.
// The noise tensor is an explicit input (the reference draws it from torch's global RNG, :29): parity needs the same
// numbers, production callers pass their own generator's output.
#include "d2s_common.h"

namespace {

// selects the top-k of v[0..d) held in LDS; returns for index i (= c0 + tid) whether it is selected and, through
// `pos_out`, how many selected indices precede it.  All 256 threads must call it (barriers inside).
__device__ __forceinline__ void topk_positions(const float* __restrict__ v, int d, int k, int* __restrict__ flag,
                                               int* __restrict__ wave_tot, int tid) {
    for (int i = tid; i < d; i += 256) {
        const float vi = v[i];
        int cnt = 0;
        for (int j = 0; j < d; ++j) {
            const float u = v[j];
            cnt += (u > vi) || (u == vi && j < i);
        }
        flag[i] = cnt < k;
    }
    __syncthreads();
    (void)wave_tot;
}

__global__ __launch_bounds__(256) void ptk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                      int* __restrict__ counts, int nS, int d, int k, float sigma,
                                                      int samples_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [d] values, [d] flags
    float* v = sh;
    int* flag = reinterpret_cast<int*>(sh + d);
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int s0 = blockIdx.x * samples_per_block, s1 = min(nS, s0 + samples_per_block);
    const float* xb = x + (long)b * d;
    int* cb = counts + (long)b * k * d;
    for (int s = s0; s < s1; ++s) {
        const float* nb = noise + ((long)b * nS + s) * d;
        for (int i = tid; i < d; i += 256) v[i] = __fadd_rn(xb[i], __fmul_rn(nb[i], sigma));
        __syncthreads();
        topk_positions(v, d, k, flag, wave_tot, tid);
        int base = 0;
        for (int c0 = 0; c0 < d; c0 += 256) {
            const int i = c0 + tid;
            const int f = (i < d) ? flag[i] : 0;
            const unsigned long long bal = __ballot(f);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += wave_tot[w];
            const int pos = base + woff + before;
            if (f && pos < k) atomicAdd(&cb[(long)pos * d + i], 1);
            base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void ptk_finalize_kernel(const int* __restrict__ counts, float* __restrict__ ind, long n, int nS) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ind[i] = (float)counts[i] / (float)nS;
}

__global__ __launch_bounds__(256) void ptk_bwd_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                      const float* __restrict__ g, float* __restrict__ gx, int nS, int d, int k,
                                                      float sigma) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float* v = sh;
    int* flag = reinterpret_cast<int*>(sh + d);
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* xb = x + (long)b * d;
    const float* gb = g + (long)b * k * d;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d <= 1024: index i = c*256 + tid
    for (int s = 0; s < nS; ++s) {
        const float* nb = noise + ((long)b * nS + s) * d;
        for (int i = tid; i < d; i += 256) v[i] = __fadd_rn(xb[i], __fmul_rn(nb[i], sigma));
        __syncthreads();
        topk_positions(v, d, k, flag, wave_tot, tid);
        int base = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int c0 = c * 256;
            if (c0 >= d) break;
            const int i = c0 + tid;
            const int f = (i < d) ? flag[i] : 0;
            const unsigned long long bal = __ballot(f);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += wave_tot[w];
            const int pos = base + woff + before;
            if (f && pos < k) acc[c] += nb[i] * gb[(long)pos * d + i];
            base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            __syncthreads();
        }
    }
    const float inv = 1.0f / (float)nS / sigma;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = c * 256 + tid;
        if (i < d) gx[(long)b * d + i] = acc[c] * inv;
    }
}

// Counter-based standard-normal generator for the perturbation noise (the reference draws torch.normal on the host and copies it to
// the device, peturbed_topk.py:29): Philox4x32-10 keyed by (seed), counter = element index / 4, four uniforms -> two Box-Muller pairs.
// Stateless and order independent: element i of the stream is the same whatever the launch shape, so a test can regenerate exactly the
// numbers a training step consumed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__global__ __launch_bounds__(256) void normal_noise_kernel(float* __restrict__ out, long n, unsigned long long seed) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;      // one Philox block = 4 outputs
    if (q * 4 >= n) return;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((unsigned long long)q >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1)
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        z[2 * h] = r * cs;
        z[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < n) out[q * 4 + j] = z[j];
}

}  // namespace

extern "C" {

// out[0..n) = standard normal numbers of the stream `seed` (device-side replacement of the reference's host torch.normal, :29)
int d2s_normal_noise(float* out, long n, unsigned long long seed, hipStream_t stream) {
    if (!out || n <= 0) return D2S_ERR_ARG;
    const long blocks = ((n + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(normal_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, n, seed);
    return d2s_check_launch();
}

size_t d2s_perturbed_topk_workspace_bytes(int b, int k, int d) { return (size_t)b * k * d * sizeof(int); }

// x [b,d], noise [b,nS,d] -> indicators [b,k,d].  workspace: b*k*d int32 counters (zeroed here).
int d2s_perturbed_topk_fwd(const float* x, const float* noise, float* indicators, int b, int nS, int d, int k, float sigma,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !noise || !indicators || b <= 0 || nS <= 0 || d <= 0 || d > 1024 || k <= 0 || k > d) return D2S_ERR_ARG;
    const size_t need = (size_t)b * k * d * sizeof(int);
    if (!workspace || workspace_bytes < need) return D2S_ERR_WORKSPACE;
    if (hipMemsetAsync(workspace, 0, need, stream) != hipSuccess) return D2S_ERR_LAUNCH;
    int spb = (nS * b + 2047) / 2048;  // aim at ~2048 workgroups
    if (spb < 1) spb = 1;
    const int chunks = (nS + spb - 1) / spb;
    hipLaunchKernelGGL(ptk_fwd_kernel, dim3(chunks, b), dim3(256), (size_t)2 * d * sizeof(float), stream, x, noise,
                       static_cast<int*>(workspace), nS, d, k, sigma, spb);
    const long n = (long)b * k * d;
    hipLaunchKernelGGL(ptk_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const int*>(workspace), indicators, n, nS);
    return d2s_check_launch();
}

// grad_out [b,k,d] -> grad_x [b,d]
int d2s_perturbed_topk_bwd(const float* x, const float* noise, const float* grad_out, float* grad_x, int b, int nS, int d, int k,
                           float sigma, hipStream_t stream) {
    if (!x || !noise || !grad_out || !grad_x || b <= 0 || nS <= 0 || d <= 0 || d > 1024 || k <= 0 || k > d) return D2S_ERR_ARG;
    hipLaunchKernelGGL(ptk_bwd_kernel, dim3(b), dim3(256), (size_t)2 * d * sizeof(float), stream, x, noise, grad_out, grad_x, nS, d,
                       k, sigma);
    return d2s_check_launch();
}

}  // extern "C"
