// Tokens-to-Token front end: soft split (nn.Unfold, vit_models/t2t_vit.py:55-57,85-99) and the
// FAVOR+ linear attention of Token_performer (vit_models/token_performer.py:31-54), forward and backward.
// Everything here is small-matrix, HBM/latency-bound work (64-wide tokens, m = 32 random features): one wave per
// token, the per-image 64x32 matrices staged in LDS, deterministic two-stage token reductions, no atomics.
#include "d2s_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// unfold: out[b, l, c*k*k + ky*k + kx] = src(b, c, oy*s - p + ky, ox*s - p + kx)   (0 outside), l = oy*Wo + ox.
// src is addressed by strides so that both an NCHW image and a token tensor [B, H*W, C] (the re-structurisation of
// t2t_vit.py:90,97, i.e. x.transpose(1,2).reshape(B,C,h,w)) are read in place.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unfold_fwd_kernel(const float* __restrict__ src, long sb, long sc, long sy, long sx,
                                                         float* __restrict__ out, int C, int H, int W, int k, int s, int p, int Ho,
                                                         int Wo, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int F = C * k * k;
    const int f = (int)(e % F);
    long r = e / F;
    const int l = (int)(r % (Ho * Wo));
    const long b = r / (Ho * Wo);
    const int c = f / (k * k), ky = (f / k) % k, kx = f % k;
    const int iy = (l / Wo) * s - p + ky, ix = (l % Wo) * s - p + kx;
    out[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? src[b * sb + c * sc + iy * sy + ix * sx] : 0.f;
}

// fold (backward of unfold): every source element gathers its <= ceil(k/s)^2 contributions in a fixed order
__global__ __launch_bounds__(256) void unfold_bwd_kernel(const float* __restrict__ g, float* __restrict__ dsrc, long sb, long sc,
                                                         long sy, long sx, int C, int H, int W, int k, int s, int p, int Ho, int Wo,
                                                         long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // enumerates (b, iy, ix, c) with c fastest
    if (e >= total) return;
    const int c = (int)(e % C);
    long r = e / C;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const long b = r / H;
    const int F = C * k * k;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ty = iy + p - ky;
        if (ty < 0 || ty % s) continue;
        const int oy = ty / s;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = ix + p - kx;
            if (tx < 0 || tx % s) continue;
            const int ox = tx / s;
            if (ox >= Wo) continue;
            acc += g[((long)b * Ho * Wo + oy * Wo + ox) * F + c * k * k + ky * k + kx];
        }
    }
    dsrc[b * sb + c * sc + iy * sy + ix * sx] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// performer, emb = 64, m = 32.  kqv rows are [k | q | v] (token_performer.py:46), row stride 192.
// ---------------------------------------------------------------------------------------------------------
constexpr int EMB = 64, MF = 32;

// kp[t,m] = exp(w_m . k_t - |k_t|^2 / 2) / sqrt(m), same for q   (token_performer.py:31-43).  One wave per token.
__global__ __launch_bounds__(256) void performer_features_fwd_kernel(const float* __restrict__ kqv, const float* __restrict__ w,
                                                                     float* __restrict__ kp, float* __restrict__ qp, long rows) {
    __shared__ float ws[MF][EMB + 1];
    __shared__ float xs[4][2][EMB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MF * EMB; i += 256) ws[i / EMB][i % EMB] = w[i];
    const long row = (long)blockIdx.x * 4 + wave;
    const bool ok = row < rows;
    float kv = 0.f, qv = 0.f;
    if (ok) { kv = kqv[row * 192 + lane]; qv = kqv[row * 192 + 64 + lane]; }
    xs[wave][0][lane] = kv;
    xs[wave][1][lane] = qv;
    const float kd = wave_sum(kv * kv) * 0.5f, qd = wave_sum(qv * qv) * 0.5f;
    __syncthreads();
    if (!ok) return;
    const int which = lane >> 5, m = lane & 31;      // lanes 0-31: k features, lanes 32-63: q features
    float dot = 0.f;
#pragma unroll 8
    for (int i = 0; i < EMB; ++i) dot += xs[wave][which][i] * ws[m][i];
    const float v = expf(dot - (which ? qd : kd)) * 0.17677669529663687f;   // 1/sqrt(32)
    (which ? qp : kp)[row * MF + m] = v;
}

// partial[b][chunk] = { A[n][m] = sum_t X[t,n] Y[t,m]  (64x32),  ysum[m] = sum_t scale_t Y[t,m] } over a chunk of tokens
__global__ __launch_bounds__(256) void token_outer_partial_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ Y,
                                                                  const float* __restrict__ scale, float* __restrict__ part, int T,
                                                                  int chunk) {
    __shared__ float ys[MF];
    __shared__ float sc;
    const int tid = threadIdx.x, n = tid & 63, mg = tid >> 6;   // thread owns A[n][mg*8 .. mg*8+7]
    const int b = blockIdx.y, t0 = blockIdx.x * chunk, t1 = min(T, t0 + chunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ysum = 0.f;
    for (int t = t0; t < t1; ++t) {
        const long row = (long)b * T + t;
        __syncthreads();
        if (tid < MF) ys[tid] = Y[row * MF + tid];
        if (tid == 0) sc = scale ? scale[row] : 1.f;
        __syncthreads();
        const float x = X[row * ldx + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += x * ys[mg * 8 + j];
        if (tid < MF) ysum += sc * ys[tid];
    }
    float* o = part + ((long)b * gridDim.x + blockIdx.x) * (EMB * MF + MF);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[n * MF + mg * 8 + j] = acc[j];
    if (tid < MF) o[EMB * MF + tid] = ysum;
}
__global__ __launch_bounds__(256) void token_outer_fold_kernel(const float* __restrict__ part, int chunks, float* __restrict__ A,
                                                               float* __restrict__ ysum) {
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= EMB * MF + MF) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[((long)b * chunks + c) * (EMB * MF + MF) + e];
    if (e < EMB * MF) A[(long)b * EMB * MF + e] = s;
    else ysum[(long)b * MF + e - EMB * MF] = s;
}

// y[t,n] = (sum_m qp[t,m] A[n,m]) / (qp_t . ksum + eps)   (token_performer.py:48-50).  One wave per token, lane = n.
__global__ __launch_bounds__(256) void performer_apply_fwd_kernel(const float* __restrict__ qp, const float* __restrict__ A,
                                                                  const float* __restrict__ ksum, float* __restrict__ y,
                                                                  float* __restrict__ Dout, int T, float eps) {
    __shared__ float As[EMB][MF + 1];
    __shared__ float ks[MF];
    __shared__ float qs[4][MF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    for (int i = tid; i < EMB * MF; i += 256) As[i / MF][i % MF] = A[(long)b * EMB * MF + i];
    if (tid < MF) ks[tid] = ksum[(long)b * MF + tid];
    const int t = blockIdx.x * 4 + wave;
    const long row = (long)b * T + t;
    if (t < T && lane < MF) qs[wave][lane] = qp[row * MF + lane];
    __syncthreads();
    if (t >= T) return;
    float d = 0.f, num = 0.f;
#pragma unroll 8
    for (int m = 0; m < MF; ++m) { d += qs[wave][m] * ks[m]; num += qs[wave][m] * As[lane][m]; }
    y[row * EMB + lane] = num / (d + eps);
    if (lane == 0) Dout[row] = d;
}

// backward of apply, per token: dnum = gy / (D+eps); dD = -(gy . y) / (D+eps); dqp[m] = sum_n dnum[n] A[n,m] + dD ksum[m]
__global__ __launch_bounds__(256) void performer_apply_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                  const float* __restrict__ Din, const float* __restrict__ A,
                                                                  const float* __restrict__ ksum, float* __restrict__ dnum,
                                                                  float* __restrict__ dD, float* __restrict__ dqp, int T, float eps) {
    __shared__ float As[EMB][MF + 1];
    __shared__ float ks[MF];
    __shared__ float dn[4][EMB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    for (int i = tid; i < EMB * MF; i += 256) As[i / MF][i % MF] = A[(long)b * EMB * MF + i];
    if (tid < MF) ks[tid] = ksum[(long)b * MF + tid];
    const int t = blockIdx.x * 4 + wave;
    const long row = (long)b * T + t;
    float g = 0.f, yy = 0.f, r = 0.f;
    if (t < T) { g = gy[row * EMB + lane]; yy = y[row * EMB + lane]; r = 1.0f / (Din[row] + eps); }
    const float dnv = g * r;
    const float dd = -wave_sum(g * yy) * r;
    dn[wave][lane] = dnv;
    __syncthreads();
    if (t >= T) return;
    dnum[row * EMB + lane] = dnv;
    if (lane == 0) dD[row] = dd;
    if (lane < MF) {
        float s = dd * ks[lane];
#pragma unroll 8
        for (int n = 0; n < EMB; ++n) s += dn[wave][n] * As[n][lane];
        dqp[row * MF + lane] = s;
    }
}

// dv[t,n] = sum_m kp[t,m] dA[n,m] (+ skip gradient);  dkp[t,m] = sum_n v[t,n] dA[n,m] + dksum[m].  dv goes to the v slice
// of dkqv (row stride 192, offset 128).
__global__ __launch_bounds__(256) void performer_kv_bwd_kernel(const float* __restrict__ kqv, const float* __restrict__ kp,
                                                               const float* __restrict__ dA, const float* __restrict__ dksum,
                                                               const float* __restrict__ skip, float* __restrict__ dkqv,
                                                               float* __restrict__ dkp, int T) {
    __shared__ float As[EMB][MF + 1];
    __shared__ float ks[MF];
    __shared__ float kps[4][MF];
    __shared__ float vs[4][EMB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    for (int i = tid; i < EMB * MF; i += 256) As[i / MF][i % MF] = dA[(long)b * EMB * MF + i];
    if (tid < MF) ks[tid] = dksum[(long)b * MF + tid];
    const int t = blockIdx.x * 4 + wave;
    const long row = (long)b * T + t;
    if (t < T) {
        vs[wave][lane] = kqv[row * 192 + 128 + lane];
        if (lane < MF) kps[wave][lane] = kp[row * MF + lane];
    }
    __syncthreads();
    if (t >= T) return;
    float dv = skip ? skip[row * EMB + lane] : 0.f;
#pragma unroll 8
    for (int m = 0; m < MF; ++m) dv += kps[wave][m] * As[lane][m];
    dkqv[row * 192 + 128 + lane] = dv;
    if (lane < MF) {
        float s = ks[lane];
#pragma unroll 8
        for (int n = 0; n < EMB; ++n) s += vs[wave][n] * As[n][lane];
        dkp[row * MF + lane] = s;
    }
}

// backward of the random features: du = dfeat * feat; dx[i] = sum_m du[m] w[m,i] - x[i] * sum_m du[m]; for k (slice 0)
// and q (slice 1) of the kqv row.
__global__ __launch_bounds__(256) void performer_features_bwd_kernel(const float* __restrict__ kqv, const float* __restrict__ w,
                                                                     const float* __restrict__ kp, const float* __restrict__ qp,
                                                                     const float* __restrict__ dkp, const float* __restrict__ dqp,
                                                                     float* __restrict__ dkqv, long rows) {
    __shared__ float ws[MF][EMB + 1];
    __shared__ float du[4][2][MF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MF * EMB; i += 256) ws[i / EMB][i % EMB] = w[i];
    const long row = (long)blockIdx.x * 4 + wave;
    const bool ok = row < rows;
    const int which = lane >> 5, m = lane & 31;
    float d = 0.f;
    if (ok) d = which ? dqp[row * MF + m] * qp[row * MF + m] : dkp[row * MF + m] * kp[row * MF + m];
    du[wave][which][m] = d;
    const float shalf = half_sum(d);     // sum over the 32 features of this half-wave
    const float sk = __shfl(shalf, 0, 64), sq = __shfl(shalf, 32, 64);
    __syncthreads();
    if (!ok) return;
    const float kx = kqv[row * 192 + lane], qx = kqv[row * 192 + 64 + lane];
    float dk = -kx * sk, dq = -qx * sq;
#pragma unroll 8
    for (int mm = 0; mm < MF; ++mm) { dk += du[wave][0][mm] * ws[mm][lane]; dq += du[wave][1][mm] * ws[mm][lane]; }
    dkqv[row * 192 + lane] = dk;
    dkqv[row * 192 + 64 + lane] = dq;
}

}  // namespace

extern "C" {

// src element (b,c,y,x) at src[b*sb + c*sc + y*sy + x*sx]; out [B, Ho*Wo, C*k*k]
int d2s_unfold_fwd(const float* src, long sb, long sc, long sy, long sx, float* out, int B, int C, int H, int W, int k, int s, int p,
                   hipStream_t stream) {
    if (!src || !out || B <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0) return D2S_ERR_ARG;
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const long total = (long)B * Ho * Wo * C * k * k;
    hipLaunchKernelGGL(unfold_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, sb, sc, sy, sx, out, C, H, W,
                       k, s, p, Ho, Wo, total);
    return d2s_check_launch();
}

int d2s_unfold_bwd(const float* g, float* dsrc, long sb, long sc, long sy, long sx, int B, int C, int H, int W, int k, int s, int p,
                   hipStream_t stream) {
    if (!g || !dsrc || B <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0) return D2S_ERR_ARG;
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(unfold_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g, dsrc, sb, sc, sy, sx, C, H, W,
                       k, s, p, Ho, Wo, total);
    return d2s_check_launch();
}

static inline int outer_chunk(int T) { return T >= 2048 ? 196 : (T >= 256 ? 64 : 32); }
size_t d2s_performer_workspace_bytes(int B, int T) {
    const int chunk = outer_chunk(T), chunks = (T + chunk - 1) / chunk;
    return (size_t)B * chunks * (EMB * MF + MF) * sizeof(float);
}

// kqv [B*T,192] = [k|q|v], w [32,64] -> y [B*T,64]; saves kp, qp [B*T,32], A [B,64,32], ksum [B,32], D [B*T]
int d2s_performer_attn_fwd(const float* kqv, const float* w, float* y, float* kp, float* qp, float* A, float* ksum, float* D, int B,
                           int T, float eps, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!kqv || !w || !y || !kp || !qp || !A || !ksum || !D || B <= 0 || T <= 0) return D2S_ERR_ARG;
    const long rows = (long)B * T;
    const int chunk = outer_chunk(T), chunks = (T + chunk - 1) / chunk;
    if (!workspace || workspace_bytes < (size_t)B * chunks * (EMB * MF + MF) * sizeof(float)) return D2S_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(performer_features_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, kqv, w, kp, qp, rows);
    hipLaunchKernelGGL(token_outer_partial_kernel, dim3(chunks, B), dim3(256), 0, stream, kqv + 128, 192L, kp, nullptr, part, T, chunk);
    hipLaunchKernelGGL(token_outer_fold_kernel, dim3((EMB * MF + MF + 255) / 256, B), dim3(256), 0, stream, part, chunks, A, ksum);
    hipLaunchKernelGGL(performer_apply_fwd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, stream, qp, A, ksum, y, D, T, eps);
    return d2s_check_launch();
}

// gy [B*T,64] (gradient of y), skip [B*T,64] or null (gradient arriving at v through the skip connection)
// -> dkqv [B*T,192] fully written.  scratch: dnum [B*T,64], dD [B*T], dqp, dkp [B*T,32], dA [B,64,32], dksum [B,32]
int d2s_performer_attn_bwd(const float* kqv, const float* w, const float* y, const float* kp, const float* qp, const float* A,
                           const float* ksum, const float* D, const float* gy, const float* skip, float* dkqv, float* dnum, float* dD,
                           float* dqp, float* dkp, float* dA, float* dksum, int B, int T, float eps, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
    if (!kqv || !w || !y || !kp || !qp || !A || !ksum || !D || !gy || !dkqv || !dnum || !dD || !dqp || !dkp || !dA || !dksum || B <= 0 ||
        T <= 0)
        return D2S_ERR_ARG;
    const long rows = (long)B * T;
    const int chunk = outer_chunk(T), chunks = (T + chunk - 1) / chunk;
    if (!workspace || workspace_bytes < (size_t)B * chunks * (EMB * MF + MF) * sizeof(float)) return D2S_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(performer_apply_bwd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, stream, gy, y, D, A, ksum, dnum, dD, dqp, T, eps);
    hipLaunchKernelGGL(token_outer_partial_kernel, dim3(chunks, B), dim3(256), 0, stream, dnum, (long)EMB, qp, dD, part, T, chunk);
    hipLaunchKernelGGL(token_outer_fold_kernel, dim3((EMB * MF + MF + 255) / 256, B), dim3(256), 0, stream, part, chunks, dA, dksum);
    hipLaunchKernelGGL(performer_kv_bwd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, stream, kqv, kp, dA, dksum, skip, dkqv, dkp, T);
    hipLaunchKernelGGL(performer_features_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, kqv, w, kp, qp, dkp, dqp,
                       dkqv, rows);
    return d2s_check_launch();
}

}  // extern "C"
