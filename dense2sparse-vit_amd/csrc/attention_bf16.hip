// Attention forward on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), head dim 64: the forward of the bf16 arithmetic mode
// (d2s_set_gemm_mode(2); BASELINE config 5's regime).  Same contract as d2s_attn_fwd_f32 - fp32 qkv in, fp32 out / log-sum-exp /
// CLS softmax row out - so the fp32 backward kernels (recompute from the saved log-sum-exp) keep working unchanged.  Q, K, V are
// rounded to bf16 on their way into registers / LDS, the scores, the softmax and both accumulations are fp32.
//
// Orientation (no cross-lane traffic for P, no shuffles for the rescale): S^T = K Q^T as in the fp32 kernel - a lane owns ONE query
// (column) and 16 of the tile's 32 keys - and then O^T = V^T P^T, so the P registers (packed to bf16) are directly the B operand and
// the output accumulator's column is again the lane's own query: the running-maximum rescale and the final 1/l are lane-local.
// The 32x32x16 MFMA wants 8 CONSECUTIVE reduction indices per lane half; the accumulator hands a lane keys {0-3, 8-11} (half 0) /
// {4-7, 12-15} (half 1) of each 16-key block.  A reduction may be walked in any order, so V^T is stored in LDS with its keys permuted
// to that order (pos(key)) and both operands agree.
#include "d2s_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int DH = 64;
constexpr int KP = 72;   // K tile row pitch in bf16 (144 B: conflict-free 16-byte fragment reads)
constexpr int VP = 40;   // V^T tile row pitch in bf16 (80 B)

__device__ __forceinline__ int key_pos(int key) {   // position of a key inside its 16-key block in the MFMA's reduction order
    const int b = key & 15;
    return (key & ~15) | (b < 4 ? b : b < 8 ? b + 4 : b < 12 ? b - 4 : b);
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 3) void attn_fwd_bf16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, float* __restrict__ cls_row, int n, int H,
                                                               float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[32 * KP];      // [key][d]
    __shared__ __attribute__((aligned(16))) __bf16 Vt[DH * VP];      // [d][pos(key)]
    extern __shared__ __attribute__((aligned(16))) float cls_s[];    // [n] raw scaled scores of query 0 (block 0 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const long ld = 3L * H * DH;
    const float* qb = qkv + (long)b * n * ld + h * DH;
    const float* kb = qb + (long)H * DH;
    const float* vb = kb + (long)H * DH;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < n;
    const bool want_cls = cls_row != nullptr && blockIdx.x == 0 && wave == 0;

    // B operand of S^T = K Q^T: this lane's query, d = 16 kk + 8 half + j, scaled, as bf16
    bf16x8 qf[4];
    {
        const int qi = min(q0 + l31, n - 1);
        const float* p = qb + (long)qi * ld + 8 * half;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + 16 * kk), c = *reinterpret_cast<const f32x4*>(p + 16 * kk + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[kk][j] = (__bf16)(a[j] * scale); qf[kk][4 + j] = (__bf16)(c[j] * scale); }
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging: thread -> (key = tid / 8, 8 consecutive d); rows past the sequence are clamped (their scores are masked below)
    const int skey = tid >> 3, sd8 = (tid & 7) * 8;
    const int ntiles = (n + 31) / 32;
    f32x4 kr[2], vr[2];
    auto fetch = [&](int t) {
        const long row = min(t * 32 + skey, n - 1);
        const float* kp = kb + row * ld + sd8;
        const float* vp = vb + row * ld + sd8;
        kr[0] = *reinterpret_cast<const f32x4*>(kp); kr[1] = *reinterpret_cast<const f32x4*>(kp + 4);
        vr[0] = *reinterpret_cast<const f32x4*>(vp); vr[1] = *reinterpret_cast<const f32x4*>(vp + 4);
    };
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        {
            bf16x8 kv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kv[j] = (__bf16)kr[0][j]; kv[4 + j] = (__bf16)kr[1][j]; }
            *reinterpret_cast<bf16x8*>(&Ks[skey * KP + sd8]) = kv;
            const int pos = key_pos(skey);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Vt[(sd8 + j) * VP + pos] = (__bf16)vr[0][j];
                Vt[(sd8 + 4 + j) * VP + pos] = (__bf16)vr[1][j];
            }
        }
        __syncthreads();
        fetch(min(t + 1, ntiles - 1));
        if (!active) continue;

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[l31 * KP + 16 * kk + 8 * half]);
            s = mfma_bf16(kf, qf[kk], s);      // s[r] = S^T[key = row(r, half)][query = l31]
        }
        const int kv0 = t * 32;
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (kv0 + mfma32_row(r, half) >= n) s[r] = -INFINITY;
            mt = fmaxf(mt, s[r]);
        }
        if (want_cls && l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + mfma32_row(r, half);
                if (key < n) cls_s[key] = s[r];
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            rs += s[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // O^T columns are this lane's own query: the rescale is lane-local
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // B operand of O^T = V^T P^T: registers r = 8 kk .. 8 kk + 7 are reduction indices 8 half .. 8 half + 7 of 16-key block kk
        bf16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[kk][j] = (__bf16)s[8 * kk + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vt[(32 * dt + l31) * VP + 16 * kk + 8 * half]);
                o[dt] = mfma_bf16(vf, pf[kk], o[dt]);      // o[dt][r] = O^T[d = 32 dt + row(r, half)][query = l31]
            }
    }
    if (!active) return;
    const bool qok = q0 + l31 < n;
    const float inv_l = 1.0f / l_run;
    if (qok) {
        float* p = out + ((long)b * n + q0 + l31) * H * DH + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // registers 4g .. 4g+3 hold d = 32 dt + 8 g + 4 half + 0..3
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv_l;
                *reinterpret_cast<f32x4*>(p + 32 * dt + 8 * g + 4 * half) = v;
            }
        if (half == 0) lse[((long)b * H + h) * n + q0 + l31] = m_run + logf(l_run);
    }
    if (want_cls) {
        const float m0 = __shfl(m_run, 0, 64), il0 = __shfl(inv_l, 0, 64);
        float* cr = cls_row + ((long)b * H + h) * n;
        for (int j = lane; j < n; j += 64) cr[j] = expf(cls_s[j] - m0) * il0;
    }
}

}  // namespace

extern "C" {

// Same contract as d2s_attn_fwd_f32; Q, K, V rounded to bf16 for the two matrix products (bf16 arithmetic mode).
int d2s_attn_fwd_bf16(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale,
                      hipStream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || n <= 0 || H <= 0 || n > 8192) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, grid, block, cls_row ? (size_t)n * sizeof(float) : 0, stream, qkv, out, lse, cls_row, n, H,
                       scale);
    return d2s_check_launch();
}

}  // extern "C"
