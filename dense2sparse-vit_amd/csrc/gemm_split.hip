// fp32-ACCURATE GEMM on the bf16 matrix cores of gfx950 ("bf16x3 split"), and with one piece a plain bf16-compute GEMM.
//
// gfx950 multiplies fp32 matrices at 157 TFLOP/s (v_mfma_f32_32x32x2_f32) but bf16 matrices at ~2.5 PFLOP/s.  Every fp32
// value is the exact sum of three bf16 numbers up to 2^-25 relative error: h = bf16(a), m = bf16(a - h), l = bf16(a - h - m)
// (8 + 8 + 8 mantissa bits; the subtractions are exact in fp32).  bf16 x bf16 products are exact in the fp32 accumulator, so
//     a*b = h_a h_b + (h_a m_b + m_a h_b) + (h_a l_b + l_a h_b + m_a m_b) + O(2^-24)
// needs 6 bf16 MFMAs per 32x32x16 block for an error of ~2^-24 per product - the same class as fp32 rounding - at an
// effective peak of 2.5 PF / 6 = 416 TFLOP/s, 2.65x the native fp32 matrix peak.  SPLIT = 1 keeps only h_a h_b: a bf16
// GEMM with fp32 accumulation (BASELINE config 5's bf16 regime).  Inputs and outputs stay fp32; the C ABI is unchanged.
//
// Two stages inside one d2s_gemm_f32 call:
//   1. split pass (HBM-bound, streaming): each operand is read once (fp32) and written as SPLIT bf16 "piece" matrices
//      [SPLIT][rows][Kp] (Kp = K rounded up to 32, zero padded).  Doing it once per operand instead of once per output tile
//      removes the 3-12x redundant conversion work of splitting inside the GEMM main loop.  For the dgrad layout (B stored
//      [K][N]) the pass transposes through LDS, so the matrix kernel only ever sees K-contiguous pieces.
//   2. matrix kernel: tile 128x128x32, 4 waves 2x2, each 64x64 = 2x2 tiles of v_mfma_f32_32x32x16_bf16; pieces are loaded with
//      16-byte loads straight into an LDS image [piece][row][40 bf16] (80-byte pitch: conflict-free ds_read_b128 fragments),
//      next K-slab prefetched to registers during the MFMAs, no VALU work in the loop; 2 workgroups per CU.
// With one piece (bf16 mode) and a large shape, stage 2 is gemm_bf16_dma_kernel instead (further down: 256x256 / 256x128 tiles fed by
// LDS-DMA), and stage 1 disappears for every operand whose producer already wrote it in bf16 (d2s_gemm_f32_bf16io: a_bf16 from
// LayerNorm / attention / a GEMM epilogue, b_bf16 from the per-step weight conversion, c_bf16 for the next GEMM).
#include "gemm_common.h"
#include <cstdlib>

namespace {
using namespace d2s_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBM = 128, SBN = 128;

__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    const float r1 = a - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// ---- flat fp32 -> bf16 copy (weights: a whole parameter arena, or one frozen weight, once; d2s_convert_bf16) ----------------------------
__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 7 < n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + i), b = *reinterpret_cast<const f32x4*>(src + i + 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = (__bf16)a[j]; v[4 + j] = (__bf16)b[j]; }
        *reinterpret_cast<bf16x8*>(dst + i) = v;
    } else {
        for (long j = i; j < n; ++j) dst[j] = (__bf16)src[j];
    }
}

// ---- stage 1a: row-major source [R][K] (ld) -> pieces [SPLIT][R][Kp] -------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, long ld, __bf16* __restrict__ dst, int R, int K,
                                                         int Kp, int vec) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per 4 consecutive k of one row
    const int kq = Kp >> 2;
    if (e >= (long)R * kq) return;
    const int row = (int)(e / kq), k = (int)(e - (long)row * kq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* p = src + (long)row * ld + k;
    if (vec && k + 3 < K) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (k + 0 < K) v[0] = p[0];
        if (k + 1 < K) v[1] = p[1];
        if (k + 2 < K) v[2] = p[2];
        if (k + 3 < K) v[3] = p[3];
    }
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 hh, mm = (__bf16)0.f, ll = (__bf16)0.f;
        if constexpr (SPLIT == 3) split3(v[j], hh, mm, ll);
        else hh = (__bf16)v[j];
        h[j] = hh; m[j] = mm; l[j] = ll;
    }
    __bf16* d = dst + (long)row * Kp + k;
    *reinterpret_cast<bf16x4*>(d) = h;
    if constexpr (SPLIT == 3) {
        *reinterpret_cast<bf16x4*>(d + (long)R * Kp) = m;
        *reinterpret_cast<bf16x4*>(d + 2L * R * Kp) = l;
    }
}

// ---- stage 1b: source [K][R] (ld, row index contiguous) -> pieces [SPLIT][R][Kp]: 64 x 64 transposes through LDS ----------------
// Loads: 16 bytes along r (4 rows of one k).  Stores: 16 bytes along k (8 bf16 of one row); a wave covers 8 rows x 128 contiguous
// bytes per instruction.  LDS image [64 k][65]: the column walk of the store phase is at worst 2-way conflicted.
template <int SPLIT, typename ST = float>
__global__ __launch_bounds__(256) void split_cols_kernel(const ST* __restrict__ src, long ld, __bf16* __restrict__ dst, int R, int K,
                                                         int Kp, int vec, float* __restrict__ colsum_part) {
    __shared__ float t[64][65];
    __shared__ float csum[4][64];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // 64 k x 16 float4
        const int f = tid + i * 256, k = f >> 4, r4 = (f & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + k < K) {
            const ST* p = src + (long)(k0 + k) * ld + r0 + r4;
            if (vec && r0 + r4 + 3 < R) {
                if constexpr (sizeof(ST) == 4) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {      // bf16 source (the bf16 data path's saved activations): 8-byte loads
                    const bf16x4 h4 = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)h4[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + r4 + j < R) v[j] = (float)p[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) t[k][r4 + j] = v[j];
    }
    __syncthreads();
    if (colsum_part) {      // exact fp32 column sums over this block's 64 k (the bias gradient of the wgrad whose dy this is), all 256
        const int c = tid & 63, q = tid >> 6;     // threads: 16 k per thread, then 4 partials per column; partial [blockIdx.y][R]
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += t[q * 16 + k][c];
        csum[q][c] = sum;
        __syncthreads();
        if (tid < 64 && r0 + tid < R) colsum_part[(long)blockIdx.y * R + r0 + tid] = (csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                 // 64 rows x 8 chunks of 8 k
        const int f = tid + i * 256, c = f & 7, r = f >> 3;
        if (r0 + r >= R || k0 + c * 8 >= Kp) continue;
        bf16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = t[c * 8 + j][r];
            __bf16 hh, mm = (__bf16)0.f, ll = (__bf16)0.f;
            if constexpr (SPLIT == 3) split3(a, hh, mm, ll);
            else hh = (__bf16)a;
            h[j] = hh; m[j] = mm; l[j] = ll;
        }
        __bf16* d = dst + (long)(r0 + r) * Kp + k0 + c * 8;
        *reinterpret_cast<bf16x8*>(d) = h;
        if constexpr (SPLIT == 3) {
            *reinterpret_cast<bf16x8*>(d + (long)R * Kp) = m;
            *reinterpret_cast<bf16x8*>(d + 2L * R * Kp) = l;
        }
    }
}

// ---- stage 2: the matrix kernel on ready-made pieces ------------------------------------------------------------------------
// Piece loads are branch-free: rows beyond the operand's extent are CLAMPED to its last row (their products land in C rows /
// columns that the epilogue never stores; K padding is real zeros written by the split pass), the per-thread byte offsets are
// computed once before the loop, and the K position is folded into a wave-uniform base pointer - so a K-step's loads are 2*SPLIT
// straight-line instructions that the scheduler can place between MFMAs.  (The earlier form had a bounds branch + zero fill per
// load: 8 exec-mask regions per K-step, issued as one block after the barrier.)
template <int BK>
__device__ __forceinline__ void piece_offsets(int rows, int Kp, int row0, int tid, unsigned (&off)[BK / 16]) {
    constexpr int CH = BK / 8;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
        const int f = tid + i * 256;
        const int row = min(row0 + f / CH, rows - 1);
        off[i] = ((unsigned)row * (unsigned)Kp + (unsigned)(f % CH) * 8u) * (unsigned)sizeof(__bf16);
    }
}
template <int SPLIT, int BK>
__device__ __forceinline__ void load_pieces(const __bf16* __restrict__ P, long piece_stride, int k0, const unsigned (&off)[BK / 16],
                                            u32x4 (&r)[SPLIT][BK / 16]) {
#pragma unroll
    for (int s = 0; s < SPLIT; ++s) {
        const char* base = reinterpret_cast<const char*>(P + s * piece_stride + k0);      // wave-uniform
#pragma unroll
        for (int i = 0; i < BK / 16; ++i) r[s][i] = *reinterpret_cast<const u32x4*>(base + off[i]);
    }
}
template <int SPLIT, int BK>
__device__ __forceinline__ void store_pieces(__bf16* __restrict__ S, int tid, const u32x4 (&r)[SPLIT][BK / 16]) {
    constexpr int CH = BK / 8, PITCH = BK + 8;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
        const int f = tid + i * 256;
        const int row = f / CH, ch = f % CH;
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) *reinterpret_cast<u32x4*>(S + (s * 128 + row) * PITCH + ch * 8) = r[s][i];
    }
}

// ---- activation operand straight from fp32 (split in registers on its way to LDS): 4 B/element of L2->CU traffic instead of
// 2*SPLIT, and no separate split pass over the big operand.  Thread item = 8 consecutive k of one row (two 16-byte loads).
template <int BK>
__device__ __forceinline__ void load_a_f32(const float* __restrict__ A, long lda, int M, int K, int row0, int k0, int tid,
                                           f32x4 (&r)[BK / 16][2]) {
    constexpr int CH = BK / 8;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
        const int f = tid + i * 256;
        const int row = row0 + f / CH, k = k0 + (f % CH) * 8;
        r[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        r[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < M) {
            const float* p = A + (long)row * lda + k;
            if (k + 7 < K) {
                r[i][0] = *reinterpret_cast<const f32x4*>(p);
                r[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k + j < K) r[i][j >> 2][j & 3] = p[j];
            }
        }
    }
}
template <int SPLIT, int BK>
__device__ __forceinline__ void store_a_split(__bf16* __restrict__ S, int tid, const f32x4 (&r)[BK / 16][2]) {
    constexpr int CH = BK / 8, PITCH = BK + 8;
#pragma unroll
    for (int i = 0; i < BK / 16; ++i) {
        const int f = tid + i * 256;
        const int row = f / CH, ch = f % CH;
        bf16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 hh, mm = (__bf16)0.f, ll = (__bf16)0.f;
            if constexpr (SPLIT == 3) split3(r[i][j >> 2][j & 3], hh, mm, ll);
            else hh = (__bf16)r[i][j >> 2][j & 3];
            h[j] = hh; m[j] = mm; l[j] = ll;
        }
        *reinterpret_cast<bf16x8*>(S + row * PITCH + ch * 8) = h;
        if constexpr (SPLIT == 3) {
            *reinterpret_cast<bf16x8*>(S + (128 + row) * PITCH + ch * 8) = m;
            *reinterpret_cast<bf16x8*>(S + (256 + row) * PITCH + ch * 8) = l;
        }
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct PieceArgs { const __bf16* Ap; const __bf16* Bp; int Kp; };

#ifdef D2S_STAMPS   // diagnostic build only: per-workgroup {entry, loop begin, loop end, exit} in 100 MHz ticks + cycles of the loop
__device__ unsigned long long g_stamps[16 * 32768];
#define STAMP(i) if (tid == 0 && blockIdx.x < 32768) g_stamps[16 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime()
#define STAMPC(i) if (tid == 0 && blockIdx.x < 32768) g_stamps[16 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime()
// phase stamp inside K-step 5 (cycles), fenced so that the scheduler keeps it between the phases it separates
#define STAMPK(i) do { __builtin_amdgcn_sched_barrier(0); if (kt == 5) { STAMPC(i); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i)
#define STAMPC(i)
#define STAMPK(i)
#endif

template <int SPLIT, int BK, bool AF32>
__global__ __launch_bounds__(256, 2) void gemm_pieces_nt_kernel(GemmArgs p, PieceArgs q) {
    constexpr int SBK = BK, PITCH = BK + 8;   // LDS row pitch in bf16: 80 B (BK 32) / 48 B (BK 16), both conflict-free for b128 reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);   // [SPLIT][128][PITCH]
    __bf16* Bs = As + SPLIT * SBM * PITCH;               // [SPLIT][128][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    STAMP(0);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbm = (p.M + SBM - 1) / SBM, nbn = (p.N + SBN - 1) / SBN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (same as the f32 kernel): consecutive tiles of one A panel share an L2
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int row0 = (bid / nbn) * SBM, col0 = (bid % nbn) * SBN;
    // K slices (wgrad): slice z covers k in [z * k_per_slice, ...) and writes its own C slab; one slice covers all of Kp otherwise
    const int kbeg = blockIdx.z * p.k_per_slice;
    const int nk = (min(q.Kp, kbeg + p.k_per_slice) - kbeg) / SBK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[AF32 ? 1 : SPLIT][BK / 16], rb[SPLIT][BK / 16];
    f32x4 fa[BK / 16][2];
    unsigned offa[BK / 16], offb[BK / 16];
    piece_offsets<BK>(p.M, q.Kp, row0, tid, offa);
    piece_offsets<BK>(p.N, q.Kp, col0, tid, offb);
    const long strideA = (long)p.M * q.Kp, strideB = (long)p.N * q.Kp;
    if constexpr (AF32) load_a_f32<BK>(p.A, p.lda, p.M, p.K, row0, kbeg, tid, fa);
    else load_pieces<SPLIT, BK>(q.Ap, strideA, kbeg, offa, ra);
    load_pieces<SPLIT, BK>(q.Bp, strideB, kbeg, offb, rb);

    STAMP(1); STAMPC(4);
    for (int kt = 0; kt < nk; ++kt) {
        STAMPK(8);
        if constexpr (AF32) store_a_split<SPLIT, BK>(As, tid, fa);
        else store_pieces<SPLIT, BK>(As, tid, ra);
        store_pieces<SPLIT, BK>(Bs, tid, rb);
        STAMPK(9);
        __syncthreads();
        STAMPK(10);
        {   // prefetch of the next K-slab; on the last trip it re-reads the last slab (harmless) so that the body has no branch and
            // the loads can be scheduled between the MFMAs below
            const int kn = kbeg + min(kt + 1, nk - 1) * SBK;
            if constexpr (AF32) load_a_f32<BK>(p.A, p.lda, p.M, p.K, row0, kn, tid, fa);
            else load_pieces<SPLIT, BK>(q.Ap, strideA, kn, offa, ra);
            load_pieces<SPLIT, BK>(q.Bp, strideB, kn, offb, rb);
        }
        STAMPK(11);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[SPLIT][2], b[SPLIT][2];
#pragma unroll
            for (int s = 0; s < SPLIT; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[s][t] = *reinterpret_cast<const bf16x8*>(As + (s * SBM + wm * 64 + t * 32 + l31) * PITCH + ks * 16 + 8 * half);
                    b[s][t] = *reinterpret_cast<const bf16x8*>(Bs + (s * SBN + wn * 64 + t * 32 + l31) * PITCH + ks * 16 + 8 * half);
                }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x16 c = acc[mt][nt];
                    if constexpr (SPLIT == 3) {   // smallest terms first
                        c = mfma_bf16(a[0][mt], b[2][nt], c);
                        c = mfma_bf16(a[2][mt], b[0][nt], c);
                        c = mfma_bf16(a[1][mt], b[1][nt], c);
                        c = mfma_bf16(a[0][mt], b[1][nt], c);
                        c = mfma_bf16(a[1][mt], b[0][nt], c);
                    }
                    acc[mt][nt] = mfma_bf16(a[0][mt], b[0][nt], c);
                }
        }
#ifndef D2S_STAMPS
        if constexpr (!AF32) {   // issue order: one global load after every NM / NL MFMAs (a block of loads ahead of the MFMAs stalls
                                 // the wave's in-order issue on the CU's 64 B/clk vector-memory path - measured 2100 of 5400 cycles)
            constexpr int NL = 2 * SPLIT * (BK / 16), NM = (BK / 16) * 4 * (SPLIT == 3 ? 6 : 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // VMEM read
            }
        }
#endif
        STAMPK(12);
        __syncthreads();
        STAMPK(13);
    }

    STAMP(2); STAMPC(5);
    float* Cb = p.C + (long)blockIdx.z * p.slab_stride;
    if (p.vec_epilogue) {   // LDS is free after the loop's final barrier; each wave uses its own staging buffer
        float* stage = reinterpret_cast<float*>(smem_raw) + wave * epi_stage_floats(2);
        store_tile_dispatch_lds<2, 2, true>(p.epi, p, Cb, acc, row0 + wm * 64, col0 + wn * 64, lane, stage);
    } else {
    const int mbase = row0 + wm * 64, nbase = col0 + wn * 64 + l31;
    switch (p.epi) {
        case EPI_BIAS: store_tile_out<EPI_BIAS, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_RELU: store_tile_out<EPI_BIAS_RELU, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_GELU: store_tile_out<EPI_BIAS_GELU, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_RESID: store_tile_out<EPI_BIAS_RESID, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_MUL_GELU_GRAD: store_tile_out<EPI_MUL_GELU_GRAD, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_MUL_RELU_MASK: store_tile_out<EPI_MUL_RELU_MASK, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_ROWADD: store_tile_out<EPI_BIAS_ROWADD, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_ACCUM: store_tile_out<EPI_ACCUM, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
        default: store_tile_out<EPI_NONE, 2, 2>(p, Cb, acc, mbase, nbase, half); break;
    }
    }
#ifdef D2S_STAMPS
    __builtin_amdgcn_s_waitcnt(0);   // stores issued and acknowledged
    STAMP(3);
#endif
}

// ---- bf16 matrix kernel, LDS-DMA form (SPLIT = 1) ----------------------------------------------------------------------------------
// The 128x128 kernel above is bound by operand delivery: the VGPR -> LDS store path (~80 B/clk/CU for ds_write_b128 against 256 B/clk for
// the fragment reads) and the L2 -> CU path (measured ceiling ~29 B/clk/CU; a 128x128 tile asks 64 B per MFMA clock).  Here
//   * both operand tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write) into a ring of NS slab buffers, NS
//     slabs requested ahead, counted s_waitcnt vmcnt(N) + raw s_barrier once per slab (a __syncthreads() would drain the queue);
//   * tiles are 256x256 (8 waves as 2x4, BK = 64: 32 B per MFMA clock, full 128-byte lines per row, 1 workgroup per CU) or 256x128
//     (4 waves as 2x2, BK = 32, 2 workgroups per CU) for shapes whose 256x256 grid would leave CUs idle; each wave owns a 128x64
//     output block = 4x2 MFMA tiles, 6 fragment reads per 8 MFMAs;
//   * the K steps are software pipelined in registers: the fragments of step g + 1 are read while the MFMAs of step g run, and when step
//     g + 1 opens a new slab the wait + barrier sit between two MFMA groups; once every wave holds the current slab's last fragments
//     the buffer goes straight back to the DMA.
// An LDS-DMA instruction writes wave-uniform base + lane * 16 B, so the LDS image is lane-linear [row][BK bf16]; the bank swizzle
// (16-byte chunk index XOR row / rows-per-256-B) is applied to the per-lane SOURCE address and again on the fragment read.
inline int gemm_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

template <int BK>
__device__ __forceinline__ int dma_swz(int row) {
    constexpr int CPR = BK / 8, RPB = 256 / (BK * 2);      // 16-byte chunks per row, rows per 256-byte bank row
    return (row / RPB) & (CPR - 1);
}

template <int N>
__device__ __forceinline__ void wait_vm_then_barrier() {      // counted wait for this wave's older LDS-DMA loads, then the workgroup barrier
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");      // "memory": fragment reads stay behind it
}

// TOK (weight gradients on the bf16 data path, both operands at hand in bf16): the operands are read where they lie, TOKEN-major -
// q.Ap = dy [K tokens][M] (row stride p.lda), q.Bp = x [K tokens][N] (p.ldb) - and no transposing pass runs first.  A slab is 64 tokens
// x 256 features per operand, kept as two [64 tokens][128 features] images of 256-byte rows with the chunk swizzle of the CDNA guide's
// dual-use image (chunk ^ ((row & 3) << 2 | (row >> 2) & 3), applied to the DMA's source address), and the MFMA fragments - 8 tokens of
// one feature per lane - come out of ds_read_b64_tr_b16 (two 4-token blocks each): 12 transposing reads per 8 MFMAs instead of 6
// ds_read_b128, nothing else in the loop changes.  WGN = 4, BK = 64 only; K % 64 == 0, M % 8 == N % 8 == 0.
__device__ __forceinline__ int tok_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
typedef __bf16 bf16x4t __attribute__((ext_vector_type(4)));

template <int WGN, int BK, int NS, bool TOK = false>
__global__ __launch_bounds__(WGN * 128, 2) void gemm_bf16_dma_kernel(GemmArgs p, PieceArgs q) {
    static_assert(!TOK || (WGN == 4 && BK == 64), "token-major form: 256x256 tiles, 64-token slabs");
    constexpr int BM = 256, BN = 64 * WGN, NW = 2 * WGN, RB = BK * 2, KS = BK / 16;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024 / NW, NB = B_BYTES / 1024 / NW;      // 1 KB wave-instructions per wave per slab
    constexpr int LPS = NA + NB;
    static_assert(NA >= 1 && NB >= 1 && KS % 2 == 0, "tile / slab shape");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    // K slices (weight gradients: the reduction runs over tokens): slice z covers k in [z * k_per_slice, ...) and writes its own C slab
    const int nslices = (q.Kp + p.k_per_slice - 1) / p.k_per_slice;
    // Persistent workgroups (grid = one residency round): a workgroup walks tiles vb = blockIdx.x, + gridDim.x, ...  The epilogue's
    // global stores are not waited for, so they drain while the next tile's slabs load and its MFMAs run - otherwise every CU stores
    // its 256 KB at the same moment (HBM-bound bursts between compute phases, measured 12.7 us of a 32 us tile at K = 768).
    for (int vb = blockIdx.x; vb < nwg * nslices; vb += gridDim.x) {
    const int slice = vb / nwg;
    int bid = vb - slice * nwg;
    const int kbeg = slice * p.k_per_slice;
    const int nk = (min(q.Kp, kbeg + p.k_per_slice) - kbeg) / BK;
    {
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int row0 = (bid / nbn) * BM, col0 = (bid % nbn) * BN;

    // per-lane source byte offsets (K slab 0) of this wave's DMA instructions; rows past the matrix edge re-read the last row (those
    // accumulator rows are never stored)
    unsigned offa[NA], offb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if constexpr (TOK) {      // 1 KB piece g = 4 token rows of feature half g >> 4; a lane brings logical chunk (lane & 15) ^ swz(row)
            const int g = wave * NA + i, row = 4 * (g & 15) + (lane >> 4), ch = (lane & 15) ^ tok_swz(row);
            offa[i] = (unsigned)row * (unsigned)(p.lda * 2) + (unsigned)(min(row0 + 128 * (g >> 4) + 8 * ch, p.M - 8) * 2);
        } else {
            const int o = (wave * NA + i) * 1024 + lane * 16, row = o / RB, ch = (o % RB) / 16;
            offa[i] = (unsigned)(min(row0 + row, p.M - 1)) * (unsigned)(q.Kp * 2) + (unsigned)((ch ^ dma_swz<BK>(row)) * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if constexpr (TOK) {
            const int g = wave * NB + i, row = 4 * (g & 15) + (lane >> 4), ch = (lane & 15) ^ tok_swz(row);
            offb[i] = (unsigned)row * (unsigned)(p.ldb * 2) + (unsigned)(min(col0 + 128 * (g >> 4) + 8 * ch, p.N - 8) * 2);
        } else {
            const int o = (wave * NB + i) * 1024 + lane * 16, row = o / RB, ch = (o % RB) / 16;
            offb[i] = (unsigned)(min(col0 + row, p.N - 1)) * (unsigned)(q.Kp * 2) + (unsigned)((ch ^ dma_swz<BK>(row)) * 16);
        }
    }
    const unsigned char* Ag = reinterpret_cast<const unsigned char*>(q.Ap);
    const unsigned char* Bg = reinterpret_cast<const unsigned char*>(q.Bp);

    auto issue = [&](int kt, int buf) {
        unsigned char* st = smem_raw + buf * STAGE;
        const unsigned kb = (unsigned)(kbeg * 2) + (unsigned)kt * RB;
        const unsigned ka = TOK ? (unsigned)(kbeg + kt * BK) * (unsigned)(p.lda * 2) : kb;      // token-major: the slab is BK token ROWS
        const unsigned kbb = TOK ? (unsigned)(kbeg + kt * BK) * (unsigned)(p.ldb * 2) : kb;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ag + offa[i] + ka),
                                             (__attribute__((address_space(3))) void*)(st + (wave * NA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bg + offb[i] + kbb),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * NB + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets inside a slab image: row * RB + ((2 * ks + half) ^ swz(row)) * 16; the swizzle depends on the row only
    int ra[4], sa[4], rb[2], sb[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int row = wm * 128 + t * 32 + l31; ra[t] = row * RB; sa[t] = dma_swz<BK>(row); }
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int row = wn * 64 + t * 32 + l31; rb[t] = A_BYTES + row * RB; sb[t] = dma_swz<BK>(row); }

    // token-major form: byte offset of this lane's address for the transposing read of (32-feature tile t, 4-token block blk) at K step 0;
    // a K step is 16 token rows = 4096 bytes further on (the swizzle does not depend on it).  Lane 4q + p of a 16-lane group addresses
    // token row q of the block, features 4p .. 4p + 3 of the group's 16.
    int ta[4][2], tb[2][2];
    if constexpr (TOK) {
        const int g2 = l31 >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int row = 8 * half + 4 * blk + tq, f = tok_swz(row);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                ta[t][blk] = wm * 16384 + 256 * row + 16 * ((4 * t + 2 * g2 + (tp >> 1)) ^ f) + 8 * (tp & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                tb[t][blk] = A_BYTES + (wn >> 1) * 16384 + 256 * row + 16 * ((8 * (wn & 1) + 4 * t + 2 * g2 + (tp >> 1)) ^ f) + 8 * (tp & 1);
        }
    }
    auto trf = [&](const unsigned char* st, int off) {
        const bf16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4t*)(st + off));
        return lo;
    };
    auto rd = [&](const unsigned char* st, int ks, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
        if constexpr (TOK) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x4t lo = trf(st, tb[t][0] + ks * 4096), hi = trf(st, tb[t][1] + ks * 4096);
#pragma unroll
                for (int j = 0; j < 4; ++j) { b[t][j] = lo[j]; b[t][4 + j] = hi[j]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x4t lo = trf(st, ta[t][0] + ks * 4096), hi = trf(st, ta[t][1] + ks * 4096);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[t][j] = lo[j]; a[t][4 + j] = hi[j]; }
            }
        } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const bf16x8*>(st + rb[t] + (((2 * ks + half) ^ sb[t]) << 4));
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const bf16x8*>(st + ra[t] + (((2 * ks + half) ^ sa[t]) << 4));
        }
    };
    auto mm_head = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[2]) { acc[0][0] = mfma_bf16(a[0], b[0], acc[0][0]); };
    auto mm_tail = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[2]) {
#pragma unroll
        for (int i = 1; i < 8; ++i) acc[i >> 1][i & 1] = mfma_bf16(a[i >> 1], b[i & 1], acc[i >> 1][i & 1]);
    };
    // sched_barrier(0) pins the phase order (left alone, hipcc sinks the fragment reads next to their MFMAs to save registers and waits for
    // each group with nothing else in flight).  Per K step: first MFMA (its fragments were read during the previous step's MFMAs) | read
    // the next step's fragments - across a slab boundary: wait + barrier, request slab kt + NS into the buffer just drained - | 7 MFMAs.
#define D2S_PIN() __builtin_amdgcn_sched_barrier(0)
    bf16x8 fa[2][4], fb[2][2];
    // token-major form, bias gradient: the workgroups of column tile 0 also sum the dy fragments they hold anyway (wave wn takes K step wn of
    // every slab, so the 32 conversions + adds per slab and wave sit beside 32 MFMAs); one partial row per K slice goes to p.colsum
    const bool do_cs = TOK && p.colsum != nullptr && col0 == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    auto cs_add = [&](const bf16x8 (&a)[4], int ks) {
        if constexpr (TOK) {
            if (do_cs && wn == ks) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) cs[t] += (float)a[t][j];
            }
        }
    };
    // steps 0 .. KS-2 of the slab in `st` (fragments of step 0 in set 0 on entry; on exit set 1 holds step KS-1)
    auto inner_steps = [&](const unsigned char* st) {
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) {
            mm_head(fa[ks & 1], fb[ks & 1]);
            D2S_PIN();
            rd(st, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
            D2S_PIN();
            mm_tail(fa[ks & 1], fb[ks & 1]);
            cs_add(fa[ks & 1], ks);
            D2S_PIN();
        }
    };

#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < nk) issue(t, t);
    if (nk >= NS) wait_vm_then_barrier<(NS - 1) * LPS>();
    else wait_vm_then_barrier<0>();
    rd(smem_raw, 0, fa[0], fb[0]);
    int buf = 0, kt = 0;
    for (; kt + NS < nk; ++kt) {          // steady state: slabs kt + 2 .. kt + NS - 1 may still be in flight at the wait
        inner_steps(smem_raw + buf * STAGE);
        wait_vm_then_barrier<(NS - 2) * LPS>();
        mm_head(fa[1], fb[1]);
        D2S_PIN();
        issue(kt + NS, buf);
        buf = (buf + 1 == NS) ? 0 : buf + 1;
        rd(smem_raw + buf * STAGE, 0, fa[0], fb[0]);
        D2S_PIN();
        mm_tail(fa[1], fb[1]);
        cs_add(fa[1], KS - 1);
        D2S_PIN();
    }
    for (; kt + 1 < nk; ++kt) {           // drain: nothing left to request
        inner_steps(smem_raw + buf * STAGE);
        wait_vm_then_barrier<0>();
        mm_head(fa[1], fb[1]);
        D2S_PIN();
        buf = (buf + 1 == NS) ? 0 : buf + 1;
        rd(smem_raw + buf * STAGE, 0, fa[0], fb[0]);
        D2S_PIN();
        mm_tail(fa[1], fb[1]);
        cs_add(fa[1], KS - 1);
        D2S_PIN();
    }
    inner_steps(smem_raw + buf * STAGE);
    mm_head(fa[1], fb[1]);
    mm_tail(fa[1], fb[1]);
    cs_add(fa[1], KS - 1);
#undef D2S_PIN
    // every wave is done with the operand images: reuse the LDS for the epilogue's wave-private staging (raw barrier: nothing to drain)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (TOK) {
        if (do_cs) {      // workgroup-uniform.  Exchange area past the epilogue's staging buffers: [wm][wn][t][lane] floats at byte 64 K
            float* csx = reinterpret_cast<float*>(smem_raw + 65536);
#pragma unroll
            for (int t = 0; t < 4; ++t) csx[((wm * 4 + wn) * 4 + t) * 64 + lane] = cs[t];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (wn == 0 && lane < 32) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += csx[((wm * 4 + w) * 4 + t) * 64 + lane] + csx[((wm * 4 + w) * 4 + t) * 64 + 32 + lane];
                    const int row = row0 + wm * 128 + t * 32 + lane;
                    if (row < p.M) p.colsum[(long)slice * p.M + row] = v;
                }
            }
        }
    }
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * epi_stage_floats(2);
    store_tile_dispatch_lds<4, 2, true>(p.epi, p, p.C ? p.C + (long)slice * p.slab_stride : p.C, acc, row0 + wm * 128, col0 + wn * 64, lane, stage);
    // the next tile's DMA overwrites the staging area: wait for every wave's staging reads (not for the global stores)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

template <int WGN, int BK, int NS, bool TOK = false>
inline void launch_dma(const GemmArgs& pv, const PieceArgs& q, hipStream_t stream) {
    constexpr int BN = 64 * WGN;
    constexpr size_t lds = (size_t)NS * (256 + BN) * BK * 2;
    static_assert(lds >= 2 * WGN * epi_stage_floats(2) * sizeof(float), "epilogue staging must fit");
    static_assert(lds <= 160 * 1024, "LDS per CU");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<WGN, BK, NS, TOK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int tiles = ((pv.M + 255) / 256) * ((pv.N + BN - 1) / BN) * ((q.Kp + pv.k_per_slice - 1) / pv.k_per_slice);
    static const int persist_env = [] { const char* e = getenv("D2S_DMA_PERSISTENT"); return e ? atoi(e) : 1; }();
    const int resident = gemm_cus() * (WGN == 4 ? 1 : 2);      // one 256x256 or two 256x128 workgroups per CU
    const int grid = (persist_env && tiles > resident) ? resident : tiles;
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<WGN, BK, NS, TOK>), dim3(grid), dim3(WGN * 128), lds, stream, pv, q);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

namespace d2s_gemm {

size_t split_workspace_bytes(int split, int M, int N, int K) {
    const size_t Kp = (size_t)((K + 31) / 32) * 32;
    return align256((size_t)split * M * Kp * sizeof(__bf16)) + align256((size_t)split * N * Kp * sizeof(__bf16));
}

// C[M,N] = epi(A[M,K] * B^T) with B given as [N][K] (b_cols = 0) or as [K][N] (b_cols = 1, the dgrad layout).
int launch_split_gemm(const GemmArgs& p, int b_cols, int split, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const int Kp = ((p.K + 31) / 32) * 32;
    if (!workspace || workspace_bytes < split_workspace_bytes(split, p.M, p.N, p.K)) return D2S_ERR_WORKSPACE;
    if ((long)p.M * Kp * 2 >= (1L << 32) || (long)p.N * Kp * 2 >= (1L << 32)) return D2S_ERR_ARG;   // per-thread byte offsets are 32-bit
    __bf16* Ap = static_cast<__bf16*>(workspace);
    __bf16* Bp = reinterpret_cast<__bf16*>(static_cast<unsigned char*>(workspace) + align256((size_t)split * p.M * Kp * sizeof(__bf16)));
    // the caller's cached bf16 form of the B operand, always [N][K] k-contiguous (for the dgrad layout: W^T), K % 32 == 0
    const bool have_b16 = p.b16 != nullptr && split == 1;
    if (have_b16) Bp = static_cast<__bf16*>(const_cast<void*>(p.b16));
    const long ea = (long)p.M * (Kp / 4), eb = (long)p.N * (Kp / 4);
    dim3 block(256);
    // The activation operand can be converted inside the matrix kernel (read as fp32 with 16-byte loads, no split pass over it) instead
    // of being pre-split.  bf16x3 (mode 1): alone, a GEMM with <= 4 column tiles runs 9-12 % faster that way and one with 9-12 column tiles
    // 4 % slower (each panel element is re-split per column tile); inside the training step pre-splitting is 0.5-1 % faster overall, so
    // it stays the default there.  bf16 (mode 2): the conversion is one instruction per two values; with at most 12 column tiles
    // (every DeiT-S GEMM) in-kernel conversion wins (+2.7 % on the step), with 18-24 (DeiT-B qkv / fc1) the fp32 re-reads lose (-2.7 %).
    // D2S_SPLIT_A_INKERNEL = 0 / 1 forces one or the other.
    static const int a_inkernel_env = [] { const char* e = getenv("D2S_SPLIT_A_INKERNEL"); return e ? atoi(e) : -1; }();
    const int col_tiles = (p.N + SBN - 1) / SBN;
    // bf16 (one piece): LDS-DMA kernel for large shapes; D2S_SPLIT_DMA = 0 off, 1 automatic tile choice, 2 / 3 force 256x128 / 256x256
    static const int dma_env = [] { const char* e = getenv("D2S_SPLIT_DMA"); return e ? atoi(e) : 1; }();
    // Without a caller-provided bf16 A the kernel needs a conversion pass over A first (6 bytes per element of A against the 2 of reading
    // it in the matrix kernel): that pays from N ~ 900 on (measured: DeiT-B fc1 / qkv yes, proj / fc2 no).
    const bool have_a16 = p.a16 != nullptr && split == 1;      // the caller's bf16 copy of A is the piece matrix (Kp == K)
    const bool dma_shape = dma_env && split == 1 && p.M >= 2048 && p.N >= 128 && epilogue_vec_ok(p) && (have_a16 || p.N >= 1024 || dma_env > 1);
    bool dma_wide = false;
    if (dma_shape) {
        if (Kp % 64 == 0 && p.N >= 256) {
            // residency rounds of each grid (256 CUs; one 256x256 or two 256x128 workgroups per CU) x relative cost of a tile
            const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256), t128 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
            const int cus = gemm_cus();
            // cost in units of one 256x256 tile time: a CU runs one 256x256 workgroup or two 256x128 ones (measured 1.17x the time of
            // one 256x256 for the pair, 0.62x when it has a single one)
            const long f128 = t128 / (2 * cus), r128 = t128 % (2 * cus);
            const double c256 = (double)((t256 + cus - 1) / cus);
            const double c128 = f128 * 1.17 + (r128 == 0 ? 0.0 : r128 <= cus ? 0.62 : 1.17);
            dma_wide = dma_env == 3 || (dma_env != 2 && c256 <= c128);
        }
    }
    if ((p.c16 || !p.C) && (!epilogue_vec_ok(p) || split != 1)) return D2S_ERR_ARG;      // the bf16 copy is written by the 16-byte epilogue only
    if (!p.A && !have_a16) return D2S_ERR_ARG;
    if (have_a16) Ap = static_cast<__bf16*>(const_cast<void*>(p.a16));
    const bool af32 = !have_a16 && p.vecA && (a_inkernel_env >= 0 ? a_inkernel_env != 0 : (split == 1 && col_tiles <= 12 && !dma_shape));
    if (split == 3) {
        if (!af32) hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)((ea + 255) / 256)), block, 0, stream, p.A, p.lda, Ap, p.M, p.K, Kp, p.vecA);
        if (b_cols) hipLaunchKernelGGL(split_cols_kernel<3>, dim3((p.N + 63) / 64, (Kp + 63) / 64), block, 0, stream, p.B, p.ldb, Bp, p.N, p.K, Kp, p.vecB, static_cast<float*>(nullptr));
        else hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)((eb + 255) / 256)), block, 0, stream, p.B, p.ldb, Bp, p.N, p.K, Kp, p.vecB);
    } else {
        if (!af32 && !have_a16) hipLaunchKernelGGL(split_rows_kernel<1>, dim3((unsigned)((ea + 255) / 256)), block, 0, stream, p.A, p.lda, Ap, p.M, p.K, Kp, p.vecA);
        if (have_b16) { /* nothing to convert */ }
        else if (b_cols) hipLaunchKernelGGL(split_cols_kernel<1>, dim3((p.N + 63) / 64, (Kp + 63) / 64), block, 0, stream, p.B, p.ldb, Bp, p.N, p.K, Kp, p.vecB, static_cast<float*>(nullptr));
        else hipLaunchKernelGGL(split_rows_kernel<1>, dim3((unsigned)((eb + 255) / 256)), block, 0, stream, p.B, p.ldb, Bp, p.N, p.K, Kp, p.vecB);
    }
    if (dma_shape && !af32) {
        GemmArgs pb = p;
        pb.vec_epilogue = 1;
        pb.k_per_slice = Kp;
        pb.slab_stride = 0;
        PieceArgs qb{Ap, Bp, Kp};
        if (dma_wide) launch_dma<4, 64, 2>(pb, qb, stream);
        else launch_dma<2, 32, 3>(pb, qb, stream);
        return d2s_check_launch();
    }
    const int tiles = ((p.M + SBM - 1) / SBM) * ((p.N + SBN - 1) / SBN);
    static const int bk = [] { const char* e = getenv("D2S_SPLIT_BK"); return (e && atoi(e) == 16) ? 16 : 32; }();   // 32 measured faster than 16 on every model shape
    size_t lds = (size_t)split * (SBM + SBN) * (bk + 8) * sizeof(__bf16);
    if (lds < 4 * epi_stage_floats(2) * sizeof(float)) lds = 4 * epi_stage_floats(2) * sizeof(float);
    GemmArgs pv = p;
    pv.vec_epilogue = epilogue_vec_ok(p) ? 1 : 0;
    pv.k_per_slice = Kp;      // one K slice over the padded reduction length
    pv.slab_stride = 0;
    PieceArgs q{Ap, Bp, Kp};
#define D2S_LAUNCH_PIECES(S, K_, F) hipLaunchKernelGGL((gemm_pieces_nt_kernel<S, K_, F>), dim3(tiles), block, lds, stream, pv, q)
    if (af32) {
        if (split == 3 && bk == 32) D2S_LAUNCH_PIECES(3, 32, true);
        else if (split == 3) D2S_LAUNCH_PIECES(3, 16, true);
        else if (bk == 32) D2S_LAUNCH_PIECES(1, 32, true);
        else D2S_LAUNCH_PIECES(1, 16, true);
    } else {
        if (split == 3 && bk == 32) D2S_LAUNCH_PIECES(3, 32, false);
        else if (split == 3) D2S_LAUNCH_PIECES(3, 16, false);
        else if (bk == 32) D2S_LAUNCH_PIECES(1, 32, false);
        else D2S_LAUNCH_PIECES(1, 16, false);
    }
#undef D2S_LAUNCH_PIECES
    return d2s_check_launch();
}

// wgrad on the bf16 matrix cores (mode 2): C slabs[z][M][N] = A^T B over K slice z, with A given as [K][M] and B as [K][N] (both
// token-major).  Both operands go through the transposing split pass, so the matrix kernel sees K-contiguous pieces as usual.
// The caller (gemm_f32.hip) combines the slabs in slab order.  Workspace: pieces only (the slabs are the caller's).
// The pieces are [rows][Kp] with Kp = K rounded up to 64 (zero padded), so that either matrix kernel can slice them.
static inline int tn_kp(int K) { return ((K + 63) / 64) * 64; }
size_t split_tn_pieces_bytes(int split, int M, int N, int K) {
    return align256((size_t)split * M * tn_kp(K) * sizeof(__bf16)) + align256((size_t)split * N * tn_kp(K) * sizeof(__bf16));
}
// 256x256 LDS-DMA tiles for weight gradients whose output fills them (every DeiT-S / DeiT-B Linear; not the predictor's narrow layers)
bool split_tn_use_dma(int M, int N, int K) {
    static const int env = [] { const char* e = getenv("D2S_SPLIT_DMA_TN"); return e ? atoi(e) : 1; }();
    const long padded = (long)((M + 255) / 256) * 256 * ((N + 255) / 256) * 256;
    return env && M >= 256 && N >= 256 && K >= 2048 && padded * 5 <= (long)M * N * 6;      // at most 20 % of the tile area past the edges
}
// K slices for the LDS-DMA form: about one 256x256 workgroup per CU, at least 8 slabs of 64 per slice
int split_tn_dma_slices(int M, int N, int K) {
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    int slices = (gemm_cus() + tiles / 2) / tiles;
    const int max_slices = tn_kp(K) / 512;
    if (slices > max_slices) slices = max_slices;
    return slices < 1 ? 1 : slices;
}

// Per-64-token partial column sums of a token-major bf16 matrix, [K / 64][R] floats, summed in the order split_cols_kernel uses (four runs
// of 16 tokens, then (0 + 1) + (2 + 3)): the bias gradient's partials when no transposing pass reads dy (token-major weight gradient).
template <typename ST>
__global__ __launch_bounds__(256) void colsum_tok_kernel(const ST* __restrict__ src, long ld, int R, float* __restrict__ colsum_part) {
    // block = 64 tokens x 256 columns; thread (q, c4): 16 tokens of 4 columns, 16 independent 16- / 8-byte loads (R % 4 == 0, ld % 4 == 0)
    __shared__ f32x4 part[4][64];
    const int q = threadIdx.x >> 6, c4 = threadIdx.x & 63, c = blockIdx.x * 256 + 4 * c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < R) {
        const ST* p = src + ((long)blockIdx.y * 64 + q * 16) * ld + c;
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (sizeof(ST) == 4) {
                v[k] = *reinterpret_cast<const f32x4*>(p + (long)k * ld);
            } else {
                const bf16x4t h = *reinterpret_cast<const bf16x4t*>(p + (long)k * ld);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[k][j] = (float)h[j];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += v[k][j];
    }
    part[q][c4] = s;
    __syncthreads();
    if (q == 0 && c < R) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (part[0][c4][j] + part[1][c4][j]) + (part[2][c4][j] + part[3][c4][j]);
        *reinterpret_cast<f32x4*>(colsum_part + (long)blockIdx.y * R + c) = o;
    }
}

// colsum_part (optional, [split_tn_colsum_partials(K)][M] floats): per-64-token partial column sums of A (= dy), produced by the split
// pass that reads dy anyway; the caller folds them in order into the bias gradient.
int split_tn_colsum_partials(int K) { return tn_kp(K) / 64; }
int launch_split_gemm_tn(const GemmArgs& p, int split, int slices, void* pieces_ws, float* colsum_part, hipStream_t stream, int* colsum_parts) {
    if (colsum_parts) *colsum_parts = split_tn_colsum_partials(p.K);      // partial rows written to colsum_part (the caller folds that many)
    const int Kp = tn_kp(p.K);
    if (split != 1) return D2S_ERR_ARG;
    if ((long)p.M * Kp * 2 >= (1L << 32) || (long)p.N * Kp * 2 >= (1L << 32)) return D2S_ERR_ARG;
    __bf16* Ap = static_cast<__bf16*>(pieces_ws);
    __bf16* Bp = reinterpret_cast<__bf16*>(static_cast<unsigned char*>(pieces_ws) + align256((size_t)split * p.M * Kp * sizeof(__bf16)));
    dim3 block(256);
    // Both operands at hand in bf16 (the bf16 data path's fc1 / qkv, and proj / fc2 where the caller kept the bf16 forms): the matrix kernel
    // reads them token-major as they lie (gemm_bf16_dma_kernel<.., TOK>) - no transposing pass, no piece matrices.  D2S_TN_TOKEN_MAJOR=0: A/B.
    static const int tok_env = [] { const char* e = getenv("D2S_TN_TOKEN_MAJOR"); return e ? atoi(e) : 1; }();
    if (tok_env && p.b16 && p.a16 && p.K % 64 == 0 && p.M % 8 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(p.b16) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.a16) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && split_tn_use_dma(p.M, p.N, p.K) &&
        epilogue_vec_ok(p) && p.slab_stride % 4 == 0 && p.k_per_slice % 64 == 0 && (long)p.K * p.lda * 2 < (1L << 32) && (long)p.K * p.ldb * 2 < (1L << 32)) {
        static const int cs_env = [] { const char* e = getenv("D2S_TN_COLSUM_INKERNEL"); return e ? atoi(e) : 1; }();
        GemmArgs pt = p;
        pt.colsum = nullptr;
        if (colsum_part && p.A)       // the fp32 gradient exists too: the bias gradient stays its exact column sum
            hipLaunchKernelGGL(colsum_tok_kernel<float>, dim3((p.M + 255) / 256, p.K / 64), block, 0, stream, p.A, p.lda, p.M, colsum_part);
        else if (colsum_part && cs_env && colsum_parts) {      // bf16-only gradient: summed by the matrix kernel itself, one partial row per K slice
            pt.colsum = colsum_part;
            *colsum_parts = slices;
        } else if (colsum_part)
            hipLaunchKernelGGL(colsum_tok_kernel<__bf16>, dim3((p.M + 255) / 256, p.K / 64), block, 0, stream, static_cast<const __bf16*>(p.b16), p.lda, p.M, colsum_part);
        pt.vec_epilogue = 1;
        pt.a16 = nullptr; pt.c16 = nullptr; pt.b16 = nullptr;
        PieceArgs qt{static_cast<const __bf16*>(p.b16), static_cast<const __bf16*>(p.a16), p.K};
        launch_dma<4, 64, 2, true>(pt, qt, stream);
        return d2s_check_launch();
    }
    if (p.b16 && !p.A)   // bf16 data path: dy handed over in bf16 ONLY (the gradient a GEMM epilogue / attention backward wrote in that form); its
                         // column sums - the bias gradient - are then sums of the bf16 values, as under torch.autocast
        hipLaunchKernelGGL((split_cols_kernel<1, __bf16>), dim3((p.M + 63) / 64, (Kp + 63) / 64), block, 0, stream, static_cast<const __bf16*>(p.b16), p.lda, Ap,
                           p.M, p.K, Kp, (int)((p.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.b16) & 7) == 0)), colsum_part);
    else
        hipLaunchKernelGGL(split_cols_kernel<1>, dim3((p.M + 63) / 64, (Kp + 63) / 64), block, 0, stream, p.A, p.lda, Ap, p.M, p.K, Kp, p.vecA, colsum_part);
    if (p.a16)      // weight gradient on the bf16 data path: the layer input x ([tokens][n_in]) was saved in bf16 only
        hipLaunchKernelGGL((split_cols_kernel<1, __bf16>), dim3((p.N + 63) / 64, (Kp + 63) / 64), block, 0, stream, static_cast<const __bf16*>(p.a16), p.ldb, Bp,
                           p.N, p.K, Kp, (int)((p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.a16) & 7) == 0)), static_cast<float*>(nullptr));
    else
        hipLaunchKernelGGL(split_cols_kernel<1>, dim3((p.N + 63) / 64, (Kp + 63) / 64), block, 0, stream, p.B, p.ldb, Bp, p.N, p.K, Kp, p.vecB, static_cast<float*>(nullptr));
    GemmArgs pv = p;                      // p.C / p.ldc / p.slab_stride / p.k_per_slice / p.epi were set by the caller
    pv.vec_epilogue = (epilogue_vec_ok(p) && (p.slab_stride % 4 == 0)) ? 1 : 0;
    pv.a16 = nullptr; pv.c16 = nullptr; pv.b16 = nullptr;
    PieceArgs q{Ap, Bp, Kp};
    if (split_tn_use_dma(p.M, p.N, p.K) && pv.vec_epilogue && p.k_per_slice % 64 == 0) {
        launch_dma<4, 64, 2>(pv, q, stream);
        return d2s_check_launch();
    }
    const int tiles = ((p.M + SBM - 1) / SBM) * ((p.N + SBN - 1) / SBN);
    size_t lds = (size_t)split * (SBM + SBN) * (32 + 8) * sizeof(__bf16);
    if (lds < 4 * epi_stage_floats(2) * sizeof(float)) lds = 4 * epi_stage_floats(2) * sizeof(float);
    hipLaunchKernelGGL((gemm_pieces_nt_kernel<1, 32, false>), dim3(tiles, 1, slices), block, lds, stream, pv, q);
    return d2s_check_launch();
}

#ifdef D2S_STAMPS
extern "C" int d2s_debug_read_stamps(unsigned long long* host_out, int n_wg) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n_wg * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace d2s_gemm

// dst[i] = bf16(src[i]), i < n: the bf16 form of a weight (or of a whole parameter arena) for d2s_gemm_f32_bf16io's b_bf16
extern "C" int d2s_convert_bf16(const float* src, void* dst, long n, hipStream_t stream) {
    if (!src || !dst || n <= 0 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return D2S_ERR_ARG;
    hipLaunchKernelGGL(convert_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, stream, src, static_cast<__bf16*>(dst), n);
    return d2s_check_launch();
}
