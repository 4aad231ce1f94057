// Attention on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), head dim 64: forward and backward of the bf16 arithmetic mode
// (d2s_set_gemm_mode(2); BASELINE config 5's regime).  Same contracts as d2s_attn_fwd_f32 / d2s_attn_bwd_f32 - fp32 qkv in, fp32 out /
// log-sum-exp / CLS softmax row / dqkv out - so either backward can run from either forward's outputs.  Q, K, V are
// rounded to bf16 on their way into registers / LDS, the scores, the softmax and both accumulations are fp32.
//
// Orientation (no cross-lane traffic for P, no shuffles for the rescale): S^T = K Q^T as in the fp32 kernel - a lane owns ONE query
// (column) and 16 of the tile's 32 keys - and then O^T = V^T P^T, so the P registers (packed to bf16) are directly the B operand and
// the output accumulator's column is again the lane's own query: the running-maximum rescale and the final 1/l are lane-local.
// The 32x32x16 MFMA wants 8 CONSECUTIVE reduction indices per lane half; the accumulator hands a lane keys {0-3, 8-11} (half 0) /
// {4-7, 12-15} (half 1) of each 16-key block.  A reduction may be walked in any order, so V^T is stored in LDS with its keys permuted
// to that order (pos(key)) and both operands agree.
#include "d2s_common.h"
#include <cstdlib>

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

// 8 consecutive values of a q / k / v row as fp32, from the fp32 qkv tensor or from its bf16 form (the qkv GEMM's c_bf16: the same
// values this kernel would round to itself, so both inputs give identical results)
__device__ __forceinline__ void load8(const float* __restrict__ p, f32x4 (&r)[2]) {
    r[0] = *reinterpret_cast<const f32x4*>(p);
    r[1] = *reinterpret_cast<const f32x4*>(p + 4);
}
__device__ __forceinline__ void load8(const __bf16* __restrict__ p, f32x4 (&r)[2]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[0][j] = (float)v[j]; r[1][j] = (float)v[4 + j]; }
}

template <typename QT>
__global__ __launch_bounds__(256, 3) void attn_fwd_bf16_kernel(const QT* __restrict__ qkv, float* __restrict__ out,
                                                               __bf16* __restrict__ out16, float* __restrict__ lse, float* __restrict__ cls_row, int n, int H,
                                                               float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[32 * KP];      // [key][d]
    __shared__ __attribute__((aligned(16))) __bf16 Vt[DH * VP];      // [d][pos(key)]
    extern __shared__ __attribute__((aligned(16))) float cls_s[];    // [n] raw scaled scores of query 0 (block 0 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH;
    const QT* qb = qkv + (long)b * n * ld + h * DH;
    const QT* kb = qb + (long)H * DH;
    const QT* vb = kb + (long)H * DH;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < n;
    const bool want_cls = cls_row != nullptr && bx == 0 && wave == 0;

    // B operand of S^T = K Q^T: this lane's query, d = 16 kk + 8 half + j, scaled, as bf16 (scale = 2^-3 for 64-wide heads: the product
    // is exact, so fp32 and bf16 qkv inputs round to the same operand).  The softmax is evaluated as 2^(s log2 e - m log2 e): one FMA and
    // one v_exp_f32 per element - the kernel is bound by its softmax VALU work (~150 vector instructions beside 8 MFMAs per 32-key tile
    // before round 3), not by the MFMAs
    const float qscale = scale;
    constexpr float L2E = 1.44269504088896340736f;
    bf16x8 qf[4];
    {
        const int qi = min(q0 + l31, n - 1);
        const QT* p = qb + (long)qi * ld + 8 * half;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 ac[2];
            load8(p + 16 * kk, ac);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[kk][j] = (__bf16)(ac[0][j] * qscale); qf[kk][4 + j] = (__bf16)(ac[1][j] * qscale); }
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
        load8(kb + row * ld + sd8, kr);
        load8(vb + row * ld + sd8, vr);
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
        if (kv0 + 32 > n) {          // only the last tile can hold keys past the sequence (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kv0 + mfma32_row(r, half) >= n) s[r] = -INFINITY;
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        if (want_cls && l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + mfma32_row(r, half);
                if (key < n) cls_s[key] = s[r];
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float m2 = m_new * L2E;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], L2E, -m2));
            rs += s[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        // O^T columns are this lane's own query: the rescale is lane-local - and after the first tiles the running maximum rarely
        // moves, so the 32 multiplies are skipped whenever no query of the wave needs them (exact: alpha == 1 for every lane then)
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L2E);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
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
        const long po = ((long)b * n + q0 + l31) * H * DH + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // registers 4g .. 4g+3 hold d = 32 dt + 8 g + 4 half + 0..3
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv_l;
                if (out) *reinterpret_cast<f32x4*>(out + po + 32 * dt + 8 * g + 4 * half) = v;
                if (out16) {      // bf16 copy for the projection GEMM of the bf16 mode (its a_bf16)
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    bf16x4_t hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (__bf16)v[j];
                    *reinterpret_cast<bf16x4_t*>(out16 + po + 32 * dt + 8 * g + 4 * half) = hv;
                }
            }
        if (half == 0) lse[((long)b * H + h) * n + q0 + l31] = m_run + logf(l_run);
    }
    if (want_cls) {
        const float m0 = __shfl(m_run, 0, 64), il0 = __shfl(inv_l, 0, 64);
        float* cr = cls_row + ((long)b * H + h) * n;
        for (int j = lane; j < n; j += 64) cr[j] = expf(cls_s[j] - m0) * il0;
    }
}

// ---- forward, 64-key tiles, V row-major with transposed reads (round 3) ---------------------------------------------------------------
// Same orientation and outputs as attn_fwd_bf16_kernel; what changes is the tile machinery around the softmax:
//   * 64 keys per tile: half the barriers, row maxima / sums / the rescale test once per 64 keys;
//   * V goes to LDS row-major [key][d] with two ds_write_b128 per thread (like K) instead of 16 scalar ds_write_b16 into a transposed image;
//     the A operand of O^T = V^T P^T (one d row, 8 consecutive keys per lane) is fetched with ds_read_b64_tr_b16, the hardware's transposing
//     read: per 16-lane group it reads a 4-key x 16-d block and hands lane i column i.  The accumulator hands a lane the keys {0-3, 8-11}
//     (half 0) / {4-7, 12-15} (half 1) of a 16-key block as reduction slots 0-7, i.e. exactly two 4-key blocks - no key permutation is
//     needed any more.  Row pitch 96 bf16 (48 dwords): the 4 rows x 2 groups of a half-wave's read start on banks 0, 8, ..., 56.
constexpr int KP2 = 72;   // K tile row pitch (bf16)
constexpr int VP2 = 96;   // V tile row pitch (bf16): 192 B = 48 dwords (conflict-free transposed reads)
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* __restrict__ base, int off0, int off1) {      // two 4-key blocks -> 8 reduction slots
    const bf16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)(base + off0));
    const bf16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4v*)(base + off1));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
    return r;
}

template <typename QT>
__global__ __launch_bounds__(256, 3) void attn_fwd_bf16_kernel64(const QT* __restrict__ qkv, float* __restrict__ out,
                                                                 __bf16* __restrict__ out16, float* __restrict__ lse, float* __restrict__ cls_row, int n, int H,
                                                                 float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[64 * KP2];      // [key][d]
    __shared__ __attribute__((aligned(16))) __bf16 Vs[64 * VP2];      // [key][d]
    extern __shared__ __attribute__((aligned(16))) float cls_s[];     // [n] raw scaled scores of query 0 (block 0 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH;
    const QT* qb = qkv + (long)b * n * ld + h * DH;
    const QT* kb = qb + (long)H * DH;
    const QT* vb = kb + (long)H * DH;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < n;
    const bool want_cls = cls_row != nullptr && bx == 0 && wave == 0;
    constexpr float L2E = 1.44269504088896340736f;

    bf16x8 qf[4];
    {
        const int qi = min(q0 + l31, n - 1);
        const QT* p = qb + (long)qi * ld + 8 * half;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 ac[2];
            load8(p + 16 * kk, ac);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[kk][j] = (__bf16)(ac[0][j] * scale); qf[kk][4 + j] = (__bf16)(ac[1][j] * scale); }
        }
    }
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging: thread -> (key = tid / 4 of the 64-key tile, 16 consecutive d); rows past the sequence are clamped (masked below)
    const int skey = tid >> 2, sd16 = (tid & 3) * 16;
    const int ntiles = (n + 63) / 64;
    f32x4 kr[4], vr[4];
    auto fetch = [&](int t) {
        const long row = min(t * 64 + skey, n - 1);
        f32x4 a[2];
        load8(kb + row * ld + sd16, a); kr[0] = a[0]; kr[1] = a[1];
        load8(kb + row * ld + sd16 + 8, a); kr[2] = a[0]; kr[3] = a[1];
        load8(vb + row * ld + sd16, a); vr[0] = a[0]; vr[1] = a[1];
        load8(vb + row * ld + sd16 + 8, a); vr[2] = a[0]; vr[3] = a[1];
    };
    // transposed-read addressing: lane j = 4 q + p of a 16-lane group supplies row q (a key), columns 4 p .. 4 p + 3 (d) of the block
    const int tj = l31 & 15, tq = tj >> 2, tp = tj & 3;
    const int v_lane = (4 * half + tq) * VP2 + 16 * (l31 >> 4) + 4 * tp;      // + 32 dt (d block) + (16 kk [+ 8]) * VP2 (key block)
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        {
            bf16x8 h0, h1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h0[j] = (__bf16)kr[0][j]; h0[4 + j] = (__bf16)kr[1][j]; h1[j] = (__bf16)kr[2][j]; h1[4 + j] = (__bf16)kr[3][j]; }
            *reinterpret_cast<bf16x8*>(&Ks[skey * KP2 + sd16]) = h0;
            *reinterpret_cast<bf16x8*>(&Ks[skey * KP2 + sd16 + 8]) = h1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h0[j] = (__bf16)vr[0][j]; h0[4 + j] = (__bf16)vr[1][j]; h1[j] = (__bf16)vr[2][j]; h1[4 + j] = (__bf16)vr[3][j]; }
            *reinterpret_cast<bf16x8*>(&Vs[skey * VP2 + sd16]) = h0;
            *reinterpret_cast<bf16x8*>(&Vs[skey * VP2 + sd16 + 8]) = h1;
        }
        __syncthreads();
        fetch(min(t + 1, ntiles - 1));
        if (!active) continue;

        f32x16 s[2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb2][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(32 * kb2 + l31) * KP2 + 16 * kk + 8 * half]);
                s[kb2] = mfma_bf16(kf, qf[kk], s[kb2]);      // s[kb2][r] = S^T[key = 32 kb2 + row(r, half)][query = l31]
            }
        }
        const int kv0 = t * 64;
        if (kv0 + 64 > n) {          // only the last tile can hold keys past the sequence (wave-uniform)
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + 32 * kb2 + mfma32_row(r, half) >= n) s[kb2][r] = -INFINITY;
        }
        float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, fmaxf(s[0][r], s[1][r]));
        if (want_cls && l31 == 0) {
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * kb2 + mfma32_row(r, half);
                    if (key < n) cls_s[key] = s[kb2][r];
                }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float m2 = m_new * L2E;
        float rs = 0.f;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kb2][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb2][r], L2E, -m2));
                rs += s[kb2][r];
            }
        rs += __shfl_xor(rs, 32, 64);
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L2E);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
        // B operand of O^T = V^T P^T: registers 8 q .. 8 q + 7 of s[kb2] are the reduction slots of 16-key block kk = 2 kb2 + q
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[kk >> 1][8 * (kk & 1) + j];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int base = v_lane + 32 * dt + 16 * kk * VP2;
                const bf16x8 vf = tr_frag(Vs, base, base + 8 * VP2);
                o[dt] = mfma_bf16(vf, pf, o[dt]);      // o[dt][r] = O^T[d = 32 dt + row(r, half)][query = l31]
            }
        }
    }
    if (!active) return;
    const bool qok = q0 + l31 < n;
    const float inv_l = 1.0f / l_run;
    if (qok) {
        const long po = ((long)b * n + q0 + l31) * H * DH + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv_l;
                if (out) *reinterpret_cast<f32x4*>(out + po + 32 * dt + 8 * g + 4 * half) = v;
                if (out16) {
                    bf16x4v hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (__bf16)v[j];
                    *reinterpret_cast<bf16x4v*>(out16 + po + 32 * dt + 8 * g + 4 * half) = hv;
                }
            }
        if (half == 0) lse[((long)b * H + h) * n + q0 + l31] = m_run + logf(l_run);
    }
    if (want_cls) {
        const float m0 = __shfl(m_run, 0, 64), il0 = __shfl(inv_l, 0, 64);
        float* cr = cls_row + ((long)b * H + h) * n;
        for (int j = lane; j < n; j += 64) cr[j] = expf(cls_s[j] - m0) * il0;
    }
}

// staging helpers shared by the backward kernels: thread -> (row = tid / 8 of the 32-row tile, 8 consecutive d)
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ base, long ld, int row0, int n, int tid, f32x4 (&r)[2]) {
    const long row = min(row0 + (tid >> 3), n - 1);          // clamped: consumers mask by index / by lse = +inf
    load8(base + row * ld + (tid & 7) * 8, r);
}
__device__ __forceinline__ void put_rows(__bf16* __restrict__ S, int tid, const f32x4 (&r)[2], float scale) {      // [row][d], pitch KP
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (__bf16)(r[0][j] * scale); v[4 + j] = (__bf16)(r[1][j] * scale); }
    *reinterpret_cast<bf16x8*>(&S[(tid >> 3) * KP + (tid & 7) * 8]) = v;
}
__device__ __forceinline__ void put_rows_t(__bf16* __restrict__ St, int tid, const f32x4 (&r)[2]) {                // [d][pos(row)], pitch VP
    const int pos = key_pos(tid >> 3), d8 = (tid & 7) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        St[(d8 + j) * VP + pos] = (__bf16)r[0][j];
        St[(d8 + 4 + j) * VP + pos] = (__bf16)r[1][j];
    }
}
template <typename T>
__device__ __forceinline__ void row_frags(const T* __restrict__ base, long ld, int row, int n, int half, float scale, bf16x8 (&f)[4]) {
    const T* p = base + (long)min(row, n - 1) * ld + 8 * half;       // this lane's own row, d = 16 kk + 8 half + j
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        f32x4 ac[2];
        load8(p + 16 * kk, ac);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[kk][j] = (__bf16)(ac[0][j] * scale); f[kk][4 + j] = (__bf16)(ac[1][j] * scale); }
    }
}
__device__ __forceinline__ void pack2(const f32x16& s, bf16x8 (&pf)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kk][j] = (__bf16)s[8 * kk + j];
}
// acc^T[dt] += T^T-tile (LDS [d][pos]) x packed registers: the O^T = V^T P^T step of the forward, reused by every backward product
__device__ __forceinline__ void mma_t(const __bf16* __restrict__ St, const bf16x8 (&pf)[2], int l31, int half, f32x16 (&acc)[2]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 tf = *reinterpret_cast<const bf16x8*>(&St[(32 * dt + l31) * VP + 16 * kk + 8 * half]);
            acc[dt] = mfma_bf16(tf, pf[kk], acc[dt]);
        }
}
// acc[rows of the LDS tile][this lane's own row] = tile (LDS [row][d]) x this lane's fragments
__device__ __forceinline__ f32x16 mma_rows(const __bf16* __restrict__ S, const bf16x8 (&f)[4], int l31, int half) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 tf = *reinterpret_cast<const bf16x8*>(&S[l31 * KP + 16 * kk + 8 * half]);
        acc = mfma_bf16(tf, f[kk], acc);
    }
    return acc;
}
// acc^T -> one row of 64 d (at element offset `off` of p, and of the optional bf16 copy p16)
__device__ __forceinline__ void store_t(const f32x16 (&acc)[2], float* __restrict__ p, __bf16* __restrict__ p16, long off, int half, float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[dt][4 * g + j] * mul;
            if (p) *reinterpret_cast<f32x4*>(p + off + 32 * dt + 8 * g + 4 * half) = v;
            if (p16) {
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4_t*>(p16 + off + 32 * dt + 8 * g + 4 * half) = hv;
            }
        }
}

// ---- backward, dQ: a lane owns one query; loop over key tiles (S^T, dP^T, dS^T lane-local, dQ^T = K^T dS^T) ------------------------
template <typename QT>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_bf16_kernel(const QT* __restrict__ qkv, const float* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dqkv, __bf16* __restrict__ dqkv16, int n, int H, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[32 * KP];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[32 * KP];
    __shared__ __attribute__((aligned(16))) __bf16 Kt[DH * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH, ldo = (long)H * DH;
    const QT* qb = qkv + (long)b * n * ld + h * DH;
    const QT* kb = qb + (long)H * DH;
    const QT* vb = kb + (long)H * DH;
    const float* dob = dout + (long)b * n * ldo + h * DH;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < n, qok = q0 + l31 < n;
    bf16x8 qf[4], dof[4];
    row_frags(qb, ld, q0 + l31, n, half, scale, qf);
    constexpr float L2E = 1.44269504088896340736f;      // p = 2^(s log2 e - lse log2 e): one FMA and one v_exp_f32 per element
    row_frags(dob, ldo, q0 + l31, n, half, 1.0f, dof);
    const float lse_i = qok ? lse[((long)b * H + h) * n + q0 + l31] * 1.44269504088896340736f : INFINITY;
    const float dl_i = qok ? delta[((long)b * H + h) * n + q0 + l31] : 0.f;
    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const int ntiles = (n + 31) / 32;
    f32x4 kr[2], vr[2];
    stage_rows(kb, ld, 0, n, tid, kr);
    stage_rows(vb, ld, 0, n, tid, vr);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        put_rows(Ks, tid, kr, 1.0f);
        put_rows(Vs, tid, vr, 1.0f);
        put_rows_t(Kt, tid, kr);
        __syncthreads();
        const int tn = min(t + 1, ntiles - 1) * 32;
        stage_rows(kb, ld, tn, n, tid, kr);
        stage_rows(vb, ld, tn, n, tid, vr);
        if (!active) continue;
        f32x16 s = mma_rows(Ks, qf, l31, half);        // scaled scores^T [key][query]
        const f32x16 dp = mma_rows(Vs, dof, l31, half);  // dP^T[key][query] = sum_d V[key][d] dO[query][d]
        const int kv0 = t * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], L2E, -lse_i)) * (dp[r] - dl_i);      // dS^T = P^T (dP^T - delta)
        if (kv0 + 32 > n) {          // keys past the sequence exist in the last tile only (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kv0 + mfma32_row(r, half) >= n) s[r] = 0.f;
        }
        bf16x8 pf[2];
        pack2(s, pf);
        mma_t(Kt, pf, l31, half, dq);                    // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]
    }
    if (active && qok) store_t(dq, dqkv, dqkv16, ((long)b * n + q0 + l31) * ld + h * DH, half, scale);
}

// ---- backward, dK / dV: a lane owns one key; loop over query tiles ------------------------------------------------------------------
template <typename QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel(const QT* __restrict__ qkv, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   float* __restrict__ dqkv, __bf16* __restrict__ dqkv16, int n, int H, float scale) {
    __shared__ __attribute__((aligned(16))) __bf16 Qs[32 * KP];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[32 * KP];
    __shared__ __attribute__((aligned(16))) __bf16 Qt[DH * VP];
    __shared__ __attribute__((aligned(16))) __bf16 Dt[DH * VP];
    __shared__ float lse_s[32], dl_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH, ldo = (long)H * DH;
    const QT* qb = qkv + (long)b * n * ld + h * DH;
    const QT* kb = qb + (long)H * DH;
    const QT* vb = kb + (long)H * DH;
    const float* dob = dout + (long)b * n * ldo + h * DH;
    const float* lse_b = lse + ((long)b * H + h) * n;
    const float* dl_b = delta + ((long)b * H + h) * n;
    const int k0 = bx * 128 + wave * 32;
    const bool active = k0 < n, kok = k0 + l31 < n;
    bf16x8 kf[4], vf[4];
    row_frags(kb, ld, k0 + l31, n, half, scale, kf);     // scaled copy: only the scores use it
    constexpr float L2E = 1.44269504088896340736f;
    row_frags(vb, ld, k0 + l31, n, half, 1.0f, vf);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const int ntiles = (n + 31) / 32;
    f32x4 qr[2], dr[2];
    float lr, dlr;
    auto fetch = [&](int row0) {
        stage_rows(qb, ld, row0, n, tid, qr);
        stage_rows(dob, ldo, row0, n, tid, dr);
        const int qi = row0 + (tid & 31), qc = min(qi, n - 1);
        const float l0 = lse_b[qc], d0 = dl_b[qc];
        lr = qi < n ? l0 * L2E : INFINITY;      // queries past the sequence: p = 2^(s log2 e - inf) = 0
        dlr = qi < n ? d0 : 0.f;
    };
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        put_rows(Qs, tid, qr, 1.0f);
        put_rows(Ds, tid, dr, 1.0f);
        put_rows_t(Qt, tid, qr);
        put_rows_t(Dt, tid, dr);
        if (tid < 32) { lse_s[tid] = lr; dl_s[tid] = dlr; }
        __syncthreads();
        fetch(min(t + 1, ntiles - 1) * 32);
        if (!active) continue;
        f32x16 s = mma_rows(Qs, kf, l31, half);          // S[query = row(r, half)][key = l31], scaled
        f32x16 dp = mma_rows(Ds, vf, l31, half);         // dP[query][key] = sum_d dO[query][d] V[key][d]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = mfma32_row(r, half);
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], L2E, -lse_s[qi]));
            s[r] = p;
            dp[r] = p * (dp[r] - dl_s[qi]);
        }
        bf16x8 pf[2], dsf[2];
        pack2(s, pf);
        pack2(dp, dsf);
        mma_t(Dt, pf, l31, half, dv);                    // dV^T[d][key] += sum_query dO[query][d] P[query][key]
        mma_t(Qt, dsf, l31, half, dk);                   // dK^T[d][key] += sum_query Q[query][d] dS[query][key]
    }
    if (active && kok) {
        const long row = ((long)b * n + k0 + l31) * ld + h * DH;
        store_t(dk, dqkv, dqkv16, row + (long)H * DH, half, scale);
        store_t(dv, dqkv, dqkv16, row + 2L * H * DH, half, 1.0f);
    }
}

}  // namespace

extern "C" int d2s_attn_delta(const float* out, const float* dout, float* delta, int B, int n, int H, hipStream_t stream);
// D2S_ATTN_BF16_T64 [1]: 1 = the 64-key-tile forward with transposed V reads for long sequences (n >= 384, or where 64-key tiles pad no more
// than 32-key tiles), 2 = always, 0 = never (the 32-key-tile kernel).  Same-call A/B: n = 577 182.4 -> 179.3 us, config 5 +0.9 %; n = 197 36.8 ->
// 39.4 us (256 instead of 224 padded keys), hence the rule.
static bool attn_t64(int n) {
    static const int mode = [] { const char* e = getenv("D2S_ATTN_BF16_T64"); return e ? atoi(e) : 1; }();
    return mode >= 2 || (mode == 1 && (n >= 384 || (n + 63) / 64 * 64 == (n + 31) / 32 * 32));
}

// Backward of the same mode (same contract as d2s_attn_bwd_f32): dqkv [B,n,3,H,64] fully written; delta_ws: [B,H,n] floats of scratch.
template <typename QT>
static int attn_bwd_bf16_impl(const QT* qkv, const float* out, const float* dout, const float* lse, float* dqkv, __bf16* dqkv16,
                              float* delta_ws, int B, int n, int H, float scale, hipStream_t stream) {
    if (!qkv || !out || !dout || !lse || (!dqkv && !dqkv16) || !delta_ws || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    const int rc = d2s_attn_delta(out, dout, delta_ws, B, n, H, stream);
    if (rc != D2S_OK) return rc;
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<QT>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, dqkv16, n, H, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<QT>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, dqkv16, n, H, scale);
    return d2s_check_launch();
}
extern "C" {

// Same contract as d2s_attn_fwd_f32; Q, K, V rounded to bf16 for the two matrix products (bf16 arithmetic mode).
int d2s_attn_fwd_bf16(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale,
                      hipStream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || n <= 0 || H <= 0 || n > 8192) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    if (attn_t64(n))
        hipLaunchKernelGGL(attn_fwd_bf16_kernel64<float>, grid, block, cls_row ? (size_t)n * sizeof(float) : 0, stream, qkv, out,
                           static_cast<__bf16*>(nullptr), lse, cls_row, n, H, scale);
    else
        hipLaunchKernelGGL(attn_fwd_bf16_kernel<float>, grid, block, cls_row ? (size_t)n * sizeof(float) : 0, stream, qkv, out,
                           static_cast<__bf16*>(nullptr), lse, cls_row, n, H, scale);
    return d2s_check_launch();
}

// The same forward on the bf16 data path: qkv may be given in bf16 (qkv_is_bf16 != 0: the c_bf16 of the qkv GEMM, same [B,n,3,H,64]
// layout - the values this kernel would round to itself, so the results are identical), and a dense [B, n, H*64] bf16 copy of the
// output is written for the projection GEMM (its a_bf16); out may be NULL in forward-only passes that consume the bf16 form alone.
int d2s_attn_fwd_bf16_bf16out(const void* qkv, int qkv_is_bf16, float* out, void* out_bf16, float* lse, float* cls_row, int B, int n, int H,
                              float scale, hipStream_t stream) {
    if (!qkv || !out_bf16 || !lse || B <= 0 || n <= 0 || H <= 0 || n > 8192) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    const size_t sh = cls_row ? (size_t)n * sizeof(float) : 0;
    if (attn_t64(n)) {
        if (qkv_is_bf16)
            hipLaunchKernelGGL(attn_fwd_bf16_kernel64<__bf16>, grid, block, sh, stream, static_cast<const __bf16*>(qkv), out,
                               static_cast<__bf16*>(out_bf16), lse, cls_row, n, H, scale);
        else
            hipLaunchKernelGGL(attn_fwd_bf16_kernel64<float>, grid, block, sh, stream, static_cast<const float*>(qkv), out,
                               static_cast<__bf16*>(out_bf16), lse, cls_row, n, H, scale);
        return d2s_check_launch();
    }
    if (qkv_is_bf16)
        hipLaunchKernelGGL(attn_fwd_bf16_kernel<__bf16>, grid, block, sh, stream, static_cast<const __bf16*>(qkv), out,
                           static_cast<__bf16*>(out_bf16), lse, cls_row, n, H, scale);
    else
        hipLaunchKernelGGL(attn_fwd_bf16_kernel<float>, grid, block, sh, stream, static_cast<const float*>(qkv), out,
                           static_cast<__bf16*>(out_bf16), lse, cls_row, n, H, scale);
    return d2s_check_launch();
}

int d2s_attn_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta_ws, int B, int n,
                      int H, float scale, hipStream_t stream) {
    return attn_bwd_bf16_impl<float>(qkv, out, dout, lse, dqkv, nullptr, delta_ws, B, n, H, scale, stream);
}
// ... on the bf16 data path: qkv optionally in bf16 (as in the forward), and a bf16 copy of dqkv (same [B,n,3,H,64] layout): the a_bf16
// of the qkv Linear's input-gradient GEMM and the dy_bf16 of its weight gradient; dqkv may be NULL when only that form is consumed
int d2s_attn_bwd_bf16_bf16out(const void* qkv, int qkv_is_bf16, const float* out, const float* dout, const float* lse, float* dqkv,
                              void* dqkv_bf16, float* delta_ws, int B, int n, int H, float scale, hipStream_t stream) {
    if (!dqkv_bf16) return D2S_ERR_ARG;
    if (qkv_is_bf16)
        return attn_bwd_bf16_impl<__bf16>(static_cast<const __bf16*>(qkv), out, dout, lse, dqkv, static_cast<__bf16*>(dqkv_bf16), delta_ws, B, n, H, scale, stream);
    return attn_bwd_bf16_impl<float>(static_cast<const float*>(qkv), out, dout, lse, dqkv, static_cast<__bf16*>(dqkv_bf16), delta_ws, B, n, H, scale, stream);
}

}  // extern "C"
