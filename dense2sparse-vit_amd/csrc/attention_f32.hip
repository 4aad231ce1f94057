// Fused multi-head attention, fp32 on the f32-input MFMA (32x32x2), head dim 64 (every model on the path).
// Replaces Attention.forward's q@k^T -> softmax -> @v (vit_models/dynamic_vit.py:218-229) without materialising
// the [B,H,n,n] matrix, emits the CLS row of the softmax that the reference returns (:234) and that the teacher's
// rows feed into MaskLoss (losses.py:76-79), and provides the backward (dq, dk, dv) by recomputation from the
// saved log-sum-exp.
//
// qkv is the raw output of the qkv Linear: [B, n, 3, H, 64] (row stride 3*H*64); out / dout are [B, n, H*64].
//
// Orientation trick (no cross-lane traffic for P): forward and dQ compute S^T = K Q^T, so a lane owns ONE query
// (column) and 16 of the tile's 32 keys (rows) in its accumulator registers - the row softmax is an in-lane
// reduction plus one exchange between the half-waves, and the accumulator registers are, as they stand, the
// A operand of the following P V (resp. dS K) product.  dK/dV use the natural orientation S = Q K^T for the same
// reason (their sums run over queries).
//
// MFMA maps (d2s_common.h): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; C reg r: row (r&3)+8(r>>2)+4(l>>5),
// col l&31.  The 64-long d reduction is walked as 32 steps; step s = 4t+j uses d = 8t + 4*half + j, so each lane
// fetches its operand as 8 float4 (t = 0..7).
#include "d2s_common.h"

namespace {

#ifdef D2S_ATTN_STAMPS   // diagnostic build only (tools/attn_stamps.py): per-workgroup phase stamps in shader cycles + 100 MHz wall ticks
__device__ unsigned long long g_attn_stamps[16 * 8192];
#define ASTAMP(i) if (threadIdx.x == 0) { const int L_ = blockIdx.y * gridDim.x + blockIdx.x; if (L_ < 8192) g_attn_stamps[16 * L_ + (i)] = __builtin_amdgcn_s_memtime(); }
#define ASTAMPR(i) if (threadIdx.x == 0) { const int L_ = blockIdx.y * gridDim.x + blockIdx.x; if (L_ < 8192) g_attn_stamps[16 * L_ + (i)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define ASTAMP(i)
#define ASTAMPR(i)
#endif

constexpr int DH = 64;
constexpr int PITCH = 68;  // floats; 16-B aligned rows, conflict-free ds_read_b128 for 16 distinct rows

__device__ __forceinline__ void load_rows_regs(const float* __restrict__ base, long ld, int row, int nrows, int half,
                                               float scale, f32x4 (&r)[8]) {
    // operand fragment of one token row for lane (row, half): r[t] = base[row][8t + 4half .. +3] * scale
    if (row < nrows) {
        const float* p = base + (long)row * ld + 4 * half;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            r[t] = *reinterpret_cast<const f32x4*>(p + 8 * t);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[t][j] *= scale;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// cooperative load of a 32-row x 64-float tile (rows row0.., zero-filled beyond nrows) by 256 threads: 2 float4 each
__device__ __forceinline__ void tile_load(const float* __restrict__ base, long ld, int row0, int nrows, int tid, f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, row = row0 + (f >> 4), d4 = (f & 15) * 4;
        r[i] = row < nrows ? *reinterpret_cast<const f32x4*>(base + (long)row * ld + d4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256;
        *reinterpret_cast<f32x4*>(&S[(f >> 4) * PITCH + (f & 15) * 4]) = r[i];
    }
}

// acc[r] += sum_d X[rowlane][d] * Yreg[d]  : A operand from an LDS tile (row = lane&31), B operand from registers
__device__ __forceinline__ void mma_lds_reg(const float* __restrict__ S, int l31, int half, const f32x4 (&y)[8], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&S[l31 * PITCH + 8 * t + 4 * half]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma32(a[j], y[t][j], acc);
    }
}
// out[dt] += X^T-style product: A operand = accumulator-layout registers p[r], B operand = LDS tile row mfma32_row(r,half)
// `groups` (wave-uniform, 1..4): only tile rows < 8 * groups carry non-zero p (the last, partial tile of a sequence): registers
// 4g..4g+3 cover rows [8g, 8g + 8), so whole groups of MFMAs whose A operand is exactly zero are skipped (bit-identical result).
__device__ __forceinline__ void mma_reg_lds(const f32x16& p, const float* __restrict__ S, int l31, int half, f32x16 (&o)[2],
                                            int groups = 4) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < groups) {
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const float* row = &S[mfma32_row(r, half) * PITCH];
                o[0] = mfma32(p[r], row[l31], o[0]);
                o[1] = mfma32(p[r], row[32 + l31], o[1]);
            }
        }
    }
}

// Transposed form of the same product: o[dt][r] = O^T[d = 32 dt + row(r, half)][query = l31].  The LDS tile row supplies the A operand
// and the accumulator-layout registers the B operand, so the output COLUMN is the lane's own query: the running-maximum rescale and the
// final 1 / l of the forward are lane-local (no cross-lane broadcast of alpha).
__device__ __forceinline__ void mma_lds_reg_t(const f32x16& p, const float* __restrict__ S, int l31, int half, f32x16 (&o)[2],
                                              int groups = 4) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < groups) {
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const float* row = &S[mfma32_row(r, half) * PITCH];
                o[0] = mfma32(row[l31], p[r], o[0]);
                o[1] = mfma32(row[32 + l31], p[r], o[1]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// POLICY: Attention.softmax_with_policy (vit_models/dynamic_vit.py:195-214) fused into the same pass:
//     mk_ij = policy[b,j] or (i == j);  e_ij = exp(S_ij - max_j S_ij) * mk_ij;  P_ij = (e_ij + eps/n) / (sum_j e_ij + eps)
// The running maximum is taken over ALL keys (masked ones included, as the reference does), masked keys add nothing to l or O, and the
// eps terms enter once at the end: O_i = (sum_j e_ij v_j + (eps/n) sum_j v_j) / (l_i + eps).  Saved for the backward: lse_i = m_i +
// log(l_i + eps) (so that e_ij / (l_i + eps) = exp(S_ij - lse_i) mk_ij) and cinv_i = (eps/n) / (l_i + eps).
// VARLEN: ragged packed batch (inference with a dynamic keep ratio, :935-949): image b owns rows cu[b] .. cu[b+1] of qkv / out, its
// lse / CLS row live at [h * total + cu[b] + i].
struct AttnFwdArgs {
    const float* qkv; float* out; float* lse; float* cls_row;
    const float* policy; float* cinv; const int* cu;
    int n, H; float scale, eps; int total;
};

template <bool POLICY, bool VARLEN>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[32 * PITCH];
    __shared__ __attribute__((aligned(16))) float Vs[32 * PITCH];
    __shared__ float pol_s[32];
    __shared__ __attribute__((aligned(16))) float vred[POLICY ? 16 * 64 : 4];
    __shared__ float vsum_s[64];
    extern __shared__ __attribute__((aligned(16))) float cls_s[];  // [n] raw scaled scores of query 0 (block 0 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int H = a.H;
    ASTAMP(0); ASTAMPR(8);
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH;
    const int row_base = VARLEN ? a.cu[b] : 0;
    const int n = VARLEN ? a.cu[b + 1] - row_base : a.n;
    if (VARLEN && bx * 128 >= n) return;                // block-uniform: this image has no queries in this tile
    const long tok0 = VARLEN ? (long)row_base : (long)b * n;       // first token row of the image
    const float* qb = a.qkv + tok0 * ld + h * DH;
    const float* kb = qb + (long)H * DH;
    const float* vb = kb + (long)H * DH;
    const float* polb = POLICY ? a.policy + (long)b * n : nullptr;
    const float scale = a.scale;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < n;
    const bool want_cls = a.cls_row != nullptr && bx == 0 && wave == 0;
    const int qi = q0 + l31;

    f32x4 qreg[8];
    load_rows_regs(qb, ld, qi, n, half, scale, qreg);

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 vacc = {0.f, 0.f, 0.f, 0.f};

    const int ntiles = (n + 31) / 32;
    f32x4 kr[2], vr[2];
    float pr = 0.f;
    tile_load(kb, ld, 0, n, tid, kr);
    tile_load(vb, ld, 0, n, tid, vr);
    if (POLICY && tid < 32) pr = tid < n ? polb[tid] : 0.f;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        tile_store(Ks, tid, kr);
        tile_store(Vs, tid, vr);
        if (POLICY) {
            if (tid < 32) pol_s[tid] = pr;
            vacc += vr[0];      // column sums of V over every key (rows beyond n were loaded as zeros)
            vacc += vr[1];
        }
        __syncthreads();
        if (t == 0) { ASTAMP(1); }
        if (t == 1) { ASTAMP(2); }
        if (t == 2) { ASTAMP(3); }
        if (t + 1 < ntiles) {
            tile_load(kb, ld, (t + 1) * 32, n, tid, kr);
            tile_load(vb, ld, (t + 1) * 32, n, tid, vr);
            if (POLICY && tid < 32) { const int kj = (t + 1) * 32 + tid; pr = kj < n ? polb[kj] : 0.f; }
        }
        if (!active) continue;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        mma_lds_reg(Ks, l31, half, qreg, s);  // s[r] = S^T[key = row(r,half)][query = l31], already scaled
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
            if (POLICY) {
                const int kr_ = mfma32_row(r, half);
                s[r] *= (kv0 + kr_ == qi) ? 1.f : pol_s[kr_];
            }
            rs += s[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // O is kept transposed (O^T[d][query]): its column is this lane's own query, so the rescale is lane-local
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[0][r] *= alpha;
            o[1][r] *= alpha;
        }
        mma_lds_reg_t(s, Vs, l31, half, o, min(4, (n - kv0 + 7) >> 3));
    }
    ASTAMP(4);
    const float c = POLICY ? a.eps / (float)n : 0.f;
    if (POLICY) {   // fold the 16 per-thread partial column sums of every column quad (threads with equal tid & 15) -> vsum_s[64]
        *reinterpret_cast<f32x4*>(&vred[(tid >> 4) * 64 + (tid & 15) * 4]) = vacc;
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += vred[g * 64 + tid];
            vsum_s[tid] = t * c;
        }
        __syncthreads();
    }
    if (!active) return;
    const float inv_l = 1.0f / (POLICY ? l_run + a.eps : l_run);
    if (qi < n) {      // registers 4g .. 4g+3 of o[dt] hold d = 32 dt + 8 g + 4 half + 0..3 of this lane's query
        float* p = a.out + (tok0 + qi) * H * DH + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float num = o[dt][4 * g + j];
                    if (POLICY) num += vsum_s[32 * dt + 8 * g + 4 * half + j];
                    v[j] = num * inv_l;
                }
                *reinterpret_cast<f32x4*>(p + 32 * dt + 8 * g + 4 * half) = v;
            }
    }
    const long stat0 = VARLEN ? (long)h * a.total + row_base : ((long)b * H + h) * n;
    if (half == 0 && qi < n) {
        if (a.lse) a.lse[stat0 + qi] = m_run + logf(POLICY ? l_run + a.eps : l_run);
        if (POLICY && a.cinv) a.cinv[stat0 + qi] = c * inv_l;
    }
    if (want_cls) {
        const float m0 = __shfl(m_run, 0, 64), il0 = __shfl(inv_l, 0, 64);
        float* cr = a.cls_row + stat0;
        // cls_s was written by lanes 0 and 32 of this wave only; same-wave LDS accesses are ordered
        for (int j = lane; j < n; j += 64) {
            float e = expf(cls_s[j] - m0);
            if (POLICY) e = e * (j == 0 ? 1.f : polb[j]) + c;
            cr[j] = e * il0;
        }
    }
    ASTAMP(5); ASTAMPR(9);
}

// delta[b,h,i] = sum_d dout[b,i,h,d] * out[b,i,h,d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                         float* __restrict__ delta, long rows, int n, int H) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / n, i = row - b * n;
    for (int h = 0; h < H; ++h) {
        const long e = row * H * DH + h * DH + lane;
        const float s = wave_sum(out[e] * dout[e]);
        if (lane == 0) delta[(b * H + h) * n + i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, dQ: one wave per 32 queries, loops over key tiles (same orientation as forward)
// ---------------------------------------------------------------------------------------------------------
// POLICY: dS_ij = exp(S_ij - lse_i) mk_ij (dP_ij - delta_i) with lse_i = m_i + log(l_i + eps); the eps/n term of P does not depend on
// S.  (The O(eps) gradient through the row maximum, which the reference's autograd carries because it does not detach the max, is
// left out: it is <= eps = 1e-6 relative.)
template <bool POLICY>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dqkv, int n, int H, float scale,
                                                          const float* __restrict__ policy) {
    __shared__ __attribute__((aligned(16))) float Ks[32 * PITCH];
    __shared__ __attribute__((aligned(16))) float Vs[32 * PITCH];
    __shared__ float pol_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH, ldo = (long)H * DH;
    const float* qb = qkv + (long)b * n * ld + h * DH;
    const float* kb = qb + (long)H * DH;
    const float* vb = kb + (long)H * DH;
    const float* dob = dout + (long)b * n * ldo + h * DH;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < n;

    f32x4 qreg[8], doreg[8];
    load_rows_regs(qb, ld, q0 + l31, n, half, scale, qreg);
    load_rows_regs(dob, ldo, q0 + l31, n, half, 1.0f, doreg);
    const bool qok = q0 + l31 < n;
    const float lse_i = qok ? lse[((long)b * H + h) * n + q0 + l31] : INFINITY;
    const float dl_i = qok ? delta[((long)b * H + h) * n + q0 + l31] : 0.f;

    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    const int ntiles = (n + 31) / 32;
    f32x4 kr[2], vr[2];
    float pr = 0.f;
    const float* polb = POLICY ? policy + (long)b * n : nullptr;
    tile_load(kb, ld, 0, n, tid, kr);
    tile_load(vb, ld, 0, n, tid, vr);
    if (POLICY && tid < 32) pr = tid < n ? polb[tid] : 0.f;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        tile_store(Ks, tid, kr);
        tile_store(Vs, tid, vr);
        if (POLICY && tid < 32) pol_s[tid] = pr;
        __syncthreads();
        if (t + 1 < ntiles) {
            tile_load(kb, ld, (t + 1) * 32, n, tid, kr);
            tile_load(vb, ld, (t + 1) * 32, n, tid, vr);
            if (POLICY && tid < 32) { const int kj = (t + 1) * 32 + tid; pr = kj < n ? polb[kj] : 0.f; }
        }
        if (!active) continue;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_lds_reg(Ks, l31, half, qreg, s);    // scaled scores^T
        mma_lds_reg(Vs, l31, half, doreg, dp);  // dP^T[key][query] = sum_d V[key][d] dO[query][d]
        const int kv0 = t * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr_ = mfma32_row(r, half);
            float p = (kv0 + kr_ < n) ? __expf(s[r] - lse_i) : 0.f;
            if (POLICY) p *= (kv0 + kr_ == q0 + l31) ? 1.f : pol_s[kr_];
            s[r] = p * (dp[r] - dl_i);          // dS^T
        }
        mma_reg_lds(s, Ks, l31, half, dq, min(4, (n - kv0 + 7) >> 3));      // dQ[query][d] += sum_key dS^T[key][query] K[key][d]
    }
    if (!active) return;
    float* dqb = dqkv + (long)b * n * ld + h * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qi = q0 + mfma32_row(r, half);
        if (qi < n) {
            float* p = dqb + (long)qi * ld;
            p[l31] = dq[0][r] * scale;
            p[32 + l31] = dq[1][r] * scale;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, dK / dV: one wave per 32 keys, loops over query tiles (natural orientation S = Q K^T)
// ---------------------------------------------------------------------------------------------------------
// POLICY: P_ij = exp(S_ij - lse_i) mk_ij + cinv_i feeds dV (the eps/n term reaches every key, masked or not); dS as in the dQ kernel.
template <bool POLICY>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int n, int H, float scale,
                                                           const float* __restrict__ policy, const float* __restrict__ cinv) {
    __shared__ __attribute__((aligned(16))) float Qs[32 * PITCH];
    __shared__ __attribute__((aligned(16))) float Ds[32 * PITCH];
    __shared__ float lse_s[32], dl_s[32], ci_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int bx, by;
    xcd_remap_2d(bx, by);      // all blocks of one head on one XCD (shared K / V / Q / dO panels stay in its L2)
    const int b = by / H, h = by % H;
    const long ld = 3L * H * DH, ldo = (long)H * DH;
    const float* qb = qkv + (long)b * n * ld + h * DH;
    const float* kb = qb + (long)H * DH;
    const float* vb = kb + (long)H * DH;
    const float* dob = dout + (long)b * n * ldo + h * DH;
    const float* lse_b = lse + ((long)b * H + h) * n;
    const float* dl_b = delta + ((long)b * H + h) * n;
    const int k0 = bx * 128 + wave * 32;
    const bool active = k0 < n;

    f32x4 kreg[8], vreg[8];
    load_rows_regs(kb, ld, k0 + l31, n, half, scale, kreg);  // scaled copy, used for the scores only
    load_rows_regs(vb, ld, k0 + l31, n, half, 1.0f, vreg);

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const int ntiles = (n + 31) / 32;
    f32x4 qr[2], dr[2];
    float lr = 0.f, dlr = 0.f, cir = 0.f;
    const float* ci_b = POLICY ? cinv + ((long)b * H + h) * n : nullptr;
    const float pol_key = (POLICY && k0 + l31 < n) ? policy[(long)b * n + k0 + l31] : 0.f;   // this lane's key
    tile_load(qb, ld, 0, n, tid, qr);
    tile_load(dob, ldo, 0, n, tid, dr);
    if (tid < 32) { lr = tid < n ? lse_b[tid] : INFINITY; dlr = tid < n ? dl_b[tid] : 0.f; if (POLICY) cir = tid < n ? ci_b[tid] : 0.f; }
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        tile_store(Qs, tid, qr);
        tile_store(Ds, tid, dr);
        if (tid < 32) { lse_s[tid] = lr; dl_s[tid] = dlr; if (POLICY) ci_s[tid] = cir; }
        __syncthreads();
        if (t + 1 < ntiles) {
            tile_load(qb, ld, (t + 1) * 32, n, tid, qr);
            tile_load(dob, ldo, (t + 1) * 32, n, tid, dr);
            if (tid < 32) {
                const int qi = (t + 1) * 32 + tid;
                lr = qi < n ? lse_b[qi] : INFINITY;
                dlr = qi < n ? dl_b[qi] : 0.f;
                if (POLICY) cir = qi < n ? ci_b[qi] : 0.f;
            }
        }
        if (!active) continue;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_lds_reg(Qs, l31, half, kreg, s);   // S[query = row(r,half)][key = l31], scaled
        mma_lds_reg(Ds, l31, half, vreg, dp);  // dP[query][key] = sum_d dO[query][d] V[key][d]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = mfma32_row(r, half);
            float p = __expf(s[r] - lse_s[qi]);  // rows beyond n carry lse = +inf -> p = 0
            if (POLICY) p *= (t * 32 + qi == k0 + l31) ? 1.f : pol_key;
            s[r] = POLICY ? p + ci_s[qi] : p;
            dp[r] = p * (dp[r] - dl_s[qi]);
        }
        const int qgroups = min(4, (n - t * 32 + 7) >> 3);          // query rows of this tile that exist
        mma_reg_lds(s, Ds, l31, half, dv, qgroups);     // dV[key][d] += sum_query P[query][key] dO[query][d]
        mma_reg_lds(dp, Qs, l31, half, dk, qgroups);    // dK[key][d] += sum_query dS[query][key] Q[query][d]
    }
    if (!active) return;
    float* dkb = dqkv + (long)b * n * ld + (long)H * DH + h * DH;
    float* dvb = dkb + (long)H * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ki = k0 + mfma32_row(r, half);
        if (ki < n) {
            float* pk = dkb + (long)ki * ld;
            float* pv = dvb + (long)ki * ld;
            pk[l31] = dk[0][r] * scale;
            pk[32 + l31] = dk[1][r] * scale;
            pv[l31] = dv[0][r];
            pv[32 + l31] = dv[1][r];
        }
    }
}

}  // namespace

#ifdef D2S_ATTN_STAMPS
extern "C" int d2s_debug_read_attn_stamps(unsigned long long* host_out, int n_wg) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), (size_t)n_wg * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

// qkv [B,n,3,H,64] -> out [B,n,H*64], lse [B,H,n]; cls_row [B,H,n] (softmax row of query 0) if non-null.
int d2s_attn_fwd_f32(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale,
                     hipStream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || n <= 0 || H <= 0 || n > 8192) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    AttnFwdArgs a{qkv, out, lse, cls_row, nullptr, nullptr, nullptr, n, H, scale, 0.f, 0};
    hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, block, cls_row ? (size_t)n * sizeof(float) : 0, stream, a);
    return d2s_check_launch();
}

// Attention with the keep policy of the dynamic-keep-ratio training path fused in (vit_models/dynamic_vit.py:195-214,216-236 with
// policy != None): policy [B,n] (1 = kept; entry 0 is the CLS token).  lse [B,H,n] = m + log(l + eps), cinv [B,H,n] = (eps/n)/(l + eps)
// are what the backward needs.
int d2s_attn_policy_fwd_f32(const float* qkv, const float* policy, float* out, float* lse, float* cinv, float* cls_row, int B, int n,
                            int H, float scale, float eps, hipStream_t stream) {
    if (!qkv || !policy || !out || !lse || !cinv || B <= 0 || n <= 0 || H <= 0 || n > 8192) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    AttnFwdArgs a{qkv, out, lse, cls_row, policy, cinv, nullptr, n, H, scale, eps, 0};
    hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, block, cls_row ? (size_t)n * sizeof(float) : 0, stream, a);
    return d2s_check_launch();
}

// Ragged packed batch (inference with a dynamic keep ratio, vit_models/dynamic_vit.py:935-949): qkv [total,3,H,64], image b = rows
// cu_seqlens[b] .. cu_seqlens[b+1]; out [total,H*64]; cls_row (optional) [H,total]: softmax row of each image's first (CLS) token.
// max_n bounds the longest image (sizes the grid; any upper bound works).  Forward only.
int d2s_attn_varlen_fwd_f32(const float* qkv, const int* cu_seqlens, float* out, float* cls_row, int B, int total, int max_n, int H,
                            float scale, hipStream_t stream) {
    if (!qkv || !cu_seqlens || !out || B <= 0 || total <= 0 || max_n <= 0 || max_n > 8192 || H <= 0) return D2S_ERR_ARG;
    dim3 grid((max_n + 127) / 128, B * H), block(256);
    AttnFwdArgs a{qkv, out, nullptr, cls_row, nullptr, nullptr, cu_seqlens, 0, H, scale, 0.f, total};
    hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, block, cls_row ? (size_t)max_n * sizeof(float) : 0, stream, a);
    return d2s_check_launch();
}

// delta[b,h,i] = sum_d dout[b,i,h,d] * out[b,i,h,d]: the row term of the softmax backward, shared by the fp32 and bf16 backward kernels
int d2s_attn_delta(const float* out, const float* dout, float* delta, int B, int n, int H, hipStream_t stream) {
    if (!out || !dout || !delta || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    const long rows = (long)B * n;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, out, dout, delta, rows, n, H);
    return d2s_check_launch();
}

// dqkv [B,n,3,H,64] is fully written.  delta_ws: [B,H,n] floats of scratch.
int d2s_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta_ws, int B,
                     int n, int H, float scale, hipStream_t stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    const long rows = (long)B * n;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, out, dout, delta_ws, rows, n, H);
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale,
                       static_cast<const float*>(nullptr));
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale,
                       static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
    return d2s_check_launch();
}

// The two halves of d2s_attn_bwd_f32 as separate entry points (delta_ws from d2s_attn_delta): dQ and dK/dV write disjoint thirds of
// dqkv and do not depend on each other, so a caller may issue them on two streams and let them share the GPU.
int d2s_attn_bwd_dq_f32(const float* qkv, const float* dout, const float* lse, const float* delta_ws, float* dqkv, int B, int n, int H,
                        float scale, hipStream_t stream) {
    if (!qkv || !dout || !lse || !dqkv || !delta_ws || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale,
                       static_cast<const float*>(nullptr));
    return d2s_check_launch();
}
int d2s_attn_bwd_dkv_f32(const float* qkv, const float* dout, const float* lse, const float* delta_ws, float* dqkv, int B, int n, int H,
                         float scale, hipStream_t stream) {
    if (!qkv || !dout || !lse || !dqkv || !delta_ws || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale,
                       static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
    return d2s_check_launch();
}

// Backward of d2s_attn_policy_fwd_f32 (lse, cinv as that call wrote them).
int d2s_attn_policy_bwd_f32(const float* qkv, const float* policy, const float* out, const float* dout, const float* lse,
                            const float* cinv, float* dqkv, float* delta_ws, int B, int n, int H, float scale, hipStream_t stream) {
    if (!qkv || !policy || !out || !dout || !lse || !cinv || !dqkv || !delta_ws || B <= 0 || n <= 0 || H <= 0) return D2S_ERR_ARG;
    const long rows = (long)B * n;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, out, dout, delta_ws, rows, n, H);
    dim3 grid((n + 127) / 128, B * H), block(256);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale, policy);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, block, 0, stream, qkv, dout, lse, delta_ws, dqkv, n, H, scale, policy, cinv);
    return d2s_check_launch();
}

}  // extern "C"
