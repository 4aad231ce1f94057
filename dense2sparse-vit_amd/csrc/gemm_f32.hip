// fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma
// chain).  Replaces the nn.Linear / Conv2d-as-GEMM calls of the reference's hot path
// (vit_models/dynamic_vit.py:169-175 Mlp, :218,231 qkv/proj, :298 patch conv, :491-531 predictor,
// :1006 head) and their autograd backward (dgrad / wgrad).
//
//   C[M,N] = op(A) * op(B)   with reduction length K
//     ALAY 0: A stored [M][K] (k contiguous)      ALAY 1: A stored [K][M] (m contiguous)
//     BLAY 0: B stored [N][K] (k contiguous)      BLAY 1: B stored [K][N] (n contiguous)
//   forward  y = x W^T      : ALAY0, BLAY0  ("NT")
//   dgrad    dx = dy W      : ALAY0, BLAY1  ("NN")
//   wgrad    dW = dy^T x    : ALAY1, BLAY1  ("TN"), reduction over the B*n token rows, split-K slabs
//
// Tile 128x128x16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator
// VGPRs).  LDS image is [k][row] with the row index XOR-swizzled by ((k>>2)&3)<<3, which makes both the
// transposing ds_write_b32 of k-contiguous operands and the ds_read_b32 fragment reads conflict-free.
// Global->register prefetch of tile t+1 is issued before the 32 MFMAs of tile t; one barrier per K-step.
#include "gemm_common.h"
#include <cstdio>
#include <cstdlib>

namespace {
using namespace d2s_gemm;

constexpr int BK = 16;

#ifdef D2S_STAMPS   // diagnostic build only: per-workgroup {entry, loop begin, loop end, exit} in 100 MHz ticks, loop cycles, hardware id
__device__ unsigned long long g_stamps_f32[8 * 65536];
#define FSTAMP(i) if (threadIdx.x == 0 && blockIdx.x < 65536 && blockIdx.z == 0) g_stamps_f32[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime()
#define FSTAMPC(i) if (threadIdx.x == 0 && blockIdx.x < 65536 && blockIdx.z == 0) g_stamps_f32[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime()
#define FSTAMPID(i) if (threadIdx.x == 0 && blockIdx.x < 65536 && blockIdx.z == 0) g_stamps_f32[8 * blockIdx.x + (i)] = \
    ((unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4)) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32)
#else
#define FSTAMP(i)
#define FSTAMPC(i)
#define FSTAMPID(i)
#endif

template <int LAY, int BR>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int row0, int k0, int rows, int kend,
                                          int vec, int tid, f32x4 (&r)[BR / 64]) {
    // BR rows x 16 k per tile = BR*4 float4, BR/64 per thread.
    // rows = valid extent of the row dimension, kend = valid extent of the reduction dimension
#pragma unroll
    for (int i = 0; i < BR / 64; ++i) {
        const int f = tid + i * 256;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (LAY == 0) {
            const int row = row0 + (f >> 2), k = k0 + (f & 3) * 4;
            if (row < rows) {
                const float* p = P + (long)row * ld + k;
                if (vec && k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (k + 0 < kend) v[0] = p[0];
                    if (k + 1 < kend) v[1] = p[1];
                    if (k + 2 < kend) v[2] = p[2];
                    if (k + 3 < kend) v[3] = p[3];
                }
            }
        } else {
            const int k = k0 + f / (BR / 4), row = row0 + (f % (BR / 4)) * 4;
            if (k < kend) {
                const float* p = P + (long)k * ld + row;
                if (vec && row + 3 < rows) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (row + 0 < rows) v[0] = p[0];
                    if (row + 1 < rows) v[1] = p[1];
                    if (row + 2 < rows) v[2] = p[2];
                    if (row + 3 < rows) v[3] = p[3];
                }
            }
        }
        r[i] = v;
    }
}

// FAST form (16-byte aligned operand, row extent a multiple of 4 where rows are contiguous, K range a multiple of BK): no guards
// in the loop.  Rows beyond the extent are CLAMPED to the last valid row / row-chunk (their products only reach C rows or columns
// that the epilogue never stores), the per-thread element offsets are computed once, the K position moves a wave-uniform base.
template <int LAY, int BR>
__device__ __forceinline__ void tile_offsets(long ld, int row0, int rows, int tid, long (&off)[BR / 64]) {
#pragma unroll
    for (int i = 0; i < BR / 64; ++i) {
        const int f = tid + i * 256;
        if (LAY == 0) off[i] = (long)min(row0 + (f >> 2), rows - 1) * ld + (f & 3) * 4;
        else off[i] = (long)(f / (BR / 4)) * ld + min(row0 + (f % (BR / 4)) * 4, rows - 4);
    }
}
template <int LAY, int BR>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ P, long ld, int k0, const long (&off)[BR / 64],
                                               f32x4 (&r)[BR / 64]) {
    const float* base = LAY == 0 ? P + k0 : P + (long)k0 * ld;      // wave-uniform
#pragma unroll
    for (int i = 0; i < BR / 64; ++i) r[i] = *reinterpret_cast<const f32x4*>(base + off[i]);
}

template <int LAY, int BR>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const f32x4 (&r)[BR / 64]) {
    // S: [BK][BR] floats, element (k,row) at k*BR + (row ^ (((k>>2)&3)<<3) ^ ((k&1)<<4)).  The (k&1)<<4 term makes the fragment reads of
    // the 16x16x4 form (lanes l and l+16 of a half-wave read rows k and k+1 of the same 16 columns) conflict-free as well
#pragma unroll
    for (int i = 0; i < BR / 64; ++i) {
        const int f = tid + i * 256;
        if (LAY == 0) {
            const int row = f >> 2, kq = f & 3;
            const int col = row ^ (kq << 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) S[(kq * 4 + j) * BR + (col ^ ((j & 1) << 4))] = r[i][j];
        } else {
            const int k = f / (BR / 4), row = (f % (BR / 4)) * 4;
            const int col = row ^ (((k >> 2) & 3) << 3) ^ ((k & 1) << 4);
            *reinterpret_cast<f32x4*>(&S[k * BR + col]) = r[i];
        }
    }
}

template <int ALAY, int BLAY, int BM, int BN, bool FAST>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 3 : 1) void gemm_f32_kernel(GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;   // 2x2 waves, each MT x NT MFMA tiles of 32x32
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* As = smem;                 // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;   // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FSTAMP(0); FSTAMPID(6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive workgroups (same XCD every 8) walk along N for one M panel so that the
    // A panel (activations) is shared through one L2.  Bijective remap for any grid size.
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int row0 = bm * BM, col0 = bn * BN;

    const int kbeg = blockIdx.z * p.k_per_slice;
    const int kend = min(p.K, kbeg + p.k_per_slice);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // wgrad (ALAY 1): the bn == 0 workgroups also sum their A (= dy) tiles over the token rows - the bias gradient - from the
    // registers the tile passes through anyway, instead of a separate pass that re-reads dy from HBM
    const bool do_colsum = (ALAY == 1) && p.colsum != nullptr && bn == 0;
    f32x4 csum[BM / 64];
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) csum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[BM / 64], rb[BN / 64];
    long offa[BM / 64], offb[BN / 64];
    if constexpr (FAST) {
        tile_offsets<ALAY, BM>(p.lda, row0, p.M, tid, offa);
        tile_offsets<BLAY, BN>(p.ldb, col0, p.N, tid, offb);
    }
    if (FAST && nk > 0) {
        load_tile_fast<ALAY, BM>(p.A, p.lda, kbeg, offa, ra);
        load_tile_fast<BLAY, BN>(p.B, p.ldb, kbeg, offb, rb);
        if (ALAY == 1 && do_colsum) {
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) csum[i] += ra[i];
        }
        store_tile<ALAY, BM>(As, tid, ra);
        store_tile<BLAY, BN>(Bs, tid, rb);
    }
    if (!FAST && nk > 0) {
        load_tile<ALAY, BM>(p.A, p.lda, row0, kbeg, p.M, kend, p.vecA, tid, ra);
        if (ALAY == 1 && do_colsum) {
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) csum[i] += ra[i];
        }
        load_tile<BLAY, BN>(p.B, p.ldb, col0, kbeg, p.N, kend, p.vecB, tid, rb);
        store_tile<ALAY, BM>(As, tid, ra);
        store_tile<BLAY, BN>(Bs, tid, rb);
    }
    __syncthreads();

    FSTAMP(1); FSTAMPC(4);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if constexpr (FAST) {   // branch-free prefetch (the last trip re-reads the last slab and discards it)
            const int kn = kbeg + min(kt + 1, nk - 1) * BK;
            load_tile_fast<ALAY, BM>(p.A, p.lda, kn, offa, ra);
            load_tile_fast<BLAY, BN>(p.B, p.ldb, kn, offb, rb);
        } else if (kt + 1 < nk) {
            load_tile<ALAY, BM>(p.A, p.lda, row0, kbeg + (kt + 1) * BK, p.M, kend, p.vecA, tid, ra);
            load_tile<BLAY, BN>(p.B, p.ldb, col0, kbeg + (kt + 1) * BK, p.N, kend, p.vecB, tid, rb);
        }
        const float* Ac = As + cur * BK * BM;
        const float* Bc = Bs + cur * BK * BN;
        // fragments are read one K-pair ahead of the MFMAs that consume them (two register sets), so the LDS round trip of pair
        // s + 1 runs under the four MFMAs of pair s
        float a[2][MT], b[2][NT];
        auto read_frags = [&](int sp, float (&af)[MT], float (&bf)[NT]) {
            const int k = 2 * sp + half;
            const int sw = (((k >> 2) & 3) << 3) ^ ((k & 1) << 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = Ac[k * BM + ((wm * WM + i * 32 + l31) ^ sw)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bc[k * BN + ((wn * WN + j * 32 + l31) ^ sw)];
        };
        read_frags(0, a[0], b[0]);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (s + 1 < BK / 2) read_frags(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(a[s & 1][i], b[s & 1][j], acc[i][j]);
        }
        if constexpr (FAST) {
            // Pinned issue order for the K-step (measured per tile shape, tools/gemm_bench.py): the global loads are spread between
            // the MFMA groups instead of one block ahead of them; for the narrower tiles the fragment reads of pair s + 1 are also
            // pinned ahead of the MFMAs of pair s (+4-6 % there, -1.5 % on 128x128 where the compiler's own placement is better).
            constexpr int NL = BM / 64 + BN / 64, NP = BK / 2, DSR = (MT + 1) / 2 + (NT + 1) / 2;
            if constexpr (BM == 128 && BN == 128) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {                                           // loads in the FIRST half of the MFMAs:
                    __builtin_amdgcn_sched_group_barrier(0x008, NP * MT * NT / (2 * NL), 0);  // the last one still has half a
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // K-step of MFMAs to land under
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NP * MT * NT / 2, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x100, DSR, 0);                     // DS read: pair 0
#pragma unroll
                for (int sp = 0; sp < NP; ++sp) {
                    if (sp + 1 < NP) __builtin_amdgcn_sched_group_barrier(0x100, DSR, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);             // MFMA
                    if (sp % (NP / NL) == NP / NL - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (spread: early placement measured -2 % here, +2 % on 128x128)
                }
            }
        }
        if (kt + 1 < nk) {
            store_tile<ALAY, BM>(As + (cur ^ 1) * BK * BM, tid, ra);
            store_tile<BLAY, BN>(Bs + (cur ^ 1) * BK * BN, tid, rb);
            if (ALAY == 1 && do_colsum) {
#pragma unroll
                for (int i = 0; i < BM / 64; ++i) csum[i] += ra[i];
            }
        }
        __syncthreads();
    }

    if (ALAY == 1 && do_colsum) {   // block-uniform.  Thread t holds rows (t % CH)*4..+3 for k rows t / CH (+ multiples of G)
        constexpr int CH = BM / 4, G = 256 / CH;
        f32x4 t = csum[0];
#pragma unroll
        for (int i = 1; i < BM / 64; ++i) t += csum[i];
        *reinterpret_cast<f32x4*>(&smem[(tid / CH) * BM + (tid % CH) * 4]) = t;
        __syncthreads();
        if (tid < BM && row0 + tid < p.M) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) s += smem[g * BM + tid];
            float* o = p.colsum + (long)blockIdx.z * p.M + row0 + tid;
            *o = (gridDim.z == 1 && p.colsum_accumulate) ? *o + s : s;
        }
        __syncthreads();
    }

    FSTAMP(2); FSTAMPC(5);
    // ---- epilogue: the (wave-uniform) epilogue kind is resolved ONCE, each kind has its own straight-line store loop ----
    float* Cb = p.C + (long)blockIdx.z * p.slab_stride;
    const int epi = (gridDim.z > 1) ? (int)EPI_NONE : p.epi;
    static_assert(4 * epi_stage_floats(NT) <= 2 * BK * (BM + BN), "epilogue staging must fit the operand buffers");
    if (p.vec_epilogue) {   // operand LDS is free after the loop's final barrier
        store_tile_dispatch_lds<MT, NT>(epi, p, Cb, acc, row0 + wm * WM, col0 + wn * WN, lane, smem + wave * epi_stage_floats(NT));
#ifdef D2S_STAMPS
        __builtin_amdgcn_s_waitcnt(0);
        FSTAMP(3);
#endif
        return;
    }
    const int mbase = row0 + wm * WM, nbase = col0 + wn * WN + l31;
    switch (epi) {
        case EPI_BIAS: store_tile_out<EPI_BIAS, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_RELU: store_tile_out<EPI_BIAS_RELU, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_GELU: store_tile_out<EPI_BIAS_GELU, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_RESID: store_tile_out<EPI_BIAS_RESID, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_MUL_GELU_GRAD: store_tile_out<EPI_MUL_GELU_GRAD, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_MUL_RELU_MASK: store_tile_out<EPI_MUL_RELU_MASK, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_BIAS_ROWADD: store_tile_out<EPI_BIAS_ROWADD, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        case EPI_ACCUM: store_tile_out<EPI_ACCUM, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
        default: store_tile_out<EPI_NONE, MT, NT>(p, Cb, acc, mbase, nbase, half); break;
    }
}

// The same GEMM on v_mfma_f32_16x16x4_f32 (same FLOP per cycle as 32x32x2 - 64 per clock and SIMD - and the same exact fp32 fma chain,
// k-ordered): a wave's 64x64 (32x32) sub-tile is 4x4 (2x2) accumulator blocks of 16x16.  Lane l feeds A[row = l & 15][k = l >> 4] and
// B[k = l >> 4][col = l & 15], so one fragment register covers FOUR k values: per 16-deep K-step the wave issues the same 32 ds_read_b32 as
// the 32x32x2 form but 64 MFMAs of 32 cycles instead of 32 of 64.  Used for the guard-free (FAST) shapes with the 16-byte epilogue; whether
// it is the default is a measured choice (the chip may hold a different clock on the two shapes, MI355X_MICROARCH.md 'DVFS give-back' 7).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int ALAY, int BLAY, int BM, int BN>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 3 : 1) void gemm_f32_kernel16(GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;   // 2x2 waves, each MT x NT MFMA blocks of 16x16
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* As = smem;
    float* Bs = smem + 2 * BK * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int row0 = bm * BM, col0 = bn * BN;
    const int kbeg = blockIdx.z * p.k_per_slice;
    const int kend = min(p.K, kbeg + p.k_per_slice);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_colsum = (ALAY == 1) && p.colsum != nullptr && bn == 0;
    f32x4 csum[BM / 64];
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) csum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[BM / 64], rb[BN / 64];
    long offa[BM / 64], offb[BN / 64];
    tile_offsets<ALAY, BM>(p.lda, row0, p.M, tid, offa);
    tile_offsets<BLAY, BN>(p.ldb, col0, p.N, tid, offb);
    if (nk > 0) {
        load_tile_fast<ALAY, BM>(p.A, p.lda, kbeg, offa, ra);
        load_tile_fast<BLAY, BN>(p.B, p.ldb, kbeg, offb, rb);
        if (ALAY == 1 && do_colsum) {
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) csum[i] += ra[i];
        }
        store_tile<ALAY, BM>(As, tid, ra);
        store_tile<BLAY, BN>(Bs, tid, rb);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int kn = kbeg + min(kt + 1, nk - 1) * BK;       // branch-free prefetch (the last trip re-reads the last slab and discards it)
        load_tile_fast<ALAY, BM>(p.A, p.lda, kn, offa, ra);
        load_tile_fast<BLAY, BN>(p.B, p.ldb, kn, offb, rb);
        const float* Ac = As + cur * BK * BM;
        const float* Bc = Bs + cur * BK * BN;
        float a[2][MT], b[2][NT];
        auto read_frags = [&](int q, float (&af)[MT], float (&bf)[NT]) {
            const int k = 4 * q + lq;
            const int sw = ((q & 3) << 3) ^ ((lq & 1) << 4);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = Ac[k * BM + ((wm * WM + i * 16 + l15) ^ sw)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bc[k * BN + ((wn * WN + j * 16 + l15) ^ sw)];
        };
        read_frags(0, a[0], b[0]);
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            if (q + 1 < BK / 4) read_frags(q + 1, a[(q + 1) & 1], b[(q + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(a[q & 1][i], b[q & 1][j], acc[i][j]);
        }
        {   // issue order of the K-step: fragment reads of quad q + 1 ahead of the MFMAs of quad q, global loads spread over the quads
            constexpr int NL = BM / 64 + BN / 64, NQ = BK / 4;
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT / 2, 0);
                if (q < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - MT * NT / 2, 0);
            }
        }
        if (kt + 1 < nk) {
            store_tile<ALAY, BM>(As + (cur ^ 1) * BK * BM, tid, ra);
            store_tile<BLAY, BN>(Bs + (cur ^ 1) * BK * BN, tid, rb);
            if (ALAY == 1 && do_colsum) {
#pragma unroll
                for (int i = 0; i < BM / 64; ++i) csum[i] += ra[i];
            }
        }
        __syncthreads();
    }

    if (ALAY == 1 && do_colsum) {
        constexpr int CH = BM / 4, G = 256 / CH;
        f32x4 t = csum[0];
#pragma unroll
        for (int i = 1; i < BM / 64; ++i) t += csum[i];
        *reinterpret_cast<f32x4*>(&smem[(tid / CH) * BM + (tid % CH) * 4]) = t;
        __syncthreads();
        if (tid < BM && row0 + tid < p.M) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) s += smem[g * BM + tid];
            float* o = p.colsum + (long)blockIdx.z * p.M + row0 + tid;
            *o = (gridDim.z == 1 && p.colsum_accumulate) ? *o + s : s;
        }
        __syncthreads();
    }
    float* Cb = p.C + (long)blockIdx.z * p.slab_stride;
    const int epi = (gridDim.z > 1) ? (int)EPI_NONE : p.epi;
    static_assert(4 * 16 * (WN + 4) <= 2 * BK * (BM + BN), "epilogue staging must fit the operand buffers");
    store_tile_dispatch_lds16<MT, NT>(epi, p, Cb, acc, row0 + wm * WM, col0 + wn * WN, lane, smem + wave * 16 * (WN + 4));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// "rk" form for k-contiguous operands (ALAY 0: activations / gradients [M][K]; BLAY 0: weights [N][K]): the LDS image of such an
// operand is [row][16 k] - exactly as it sits in memory - so a global float4 goes to LDS with ONE ds_write_b128 (the [k][row] image
// needs four transposing ds_write_b32) and a lane fetches its 8 k-values of a K-step with TWO ds_read_b128 (instead of eight
// ds_read_b32).  For that the k order inside a K-step is permuted: MFMA step s of a 16-deep K-step multiplies k = s (lanes 0-31) and
// k = 8 + s (lanes 32-63); the same permutation is applied to both operands, so every product a_k * b_k is still formed exactly once -
// only the order of the fp32 accumulation chain changes (still deterministic).  Per wave and K-step of the 128x128 NT tile: 8
// ds_read_b128 + 4 ds_write_b128 where the [k][row] image issues 32 ds_read_b32 + 16 ds_write_b32, for the same 32 MFMAs.
// 16-byte chunks of a row are XOR-swizzled by (row >> 2) & 3: conflict-free ds_write_b128 (4 rows x 4 chunks per 16 lanes) and
// ds_read_b128 (16 consecutive rows, one chunk).  A k-major B operand (NN layout) keeps the [k][row] image and is read with k = 8h + s.
template <int BR>
__device__ __forceinline__ void store_tile_rk(float* __restrict__ S, int tid, const f32x4 (&r)[BR / 64]) {
#pragma unroll
    for (int i = 0; i < BR / 64; ++i) {
        const int f = tid + i * 256;
        const int row = f >> 2, kq = f & 3;
        *reinterpret_cast<f32x4*>(&S[row * BK + ((kq ^ ((row >> 2) & 3)) << 2)]) = r[i];
    }
}

template <int BLAY, int BM, int BN>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 3 : 1) void gemm_f32_rk_kernel(GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* As = smem;                 // [2][BM][BK]  (rk image)
    float* Bs = smem + 2 * BK * BM;   // [2][BN][BK]  (rk image, BLAY 0)  or  [2][BK][BN]  ([k][row] image, BLAY 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int row0 = bm * BM, col0 = bn * BN;
    const int kbeg = blockIdx.z * p.k_per_slice;
    const int kend = min(p.K, kbeg + p.k_per_slice);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[BM / 64], rb[BN / 64];
    long offa[BM / 64], offb[BN / 64];
    tile_offsets<0, BM>(p.lda, row0, p.M, tid, offa);
    tile_offsets<BLAY, BN>(p.ldb, col0, p.N, tid, offb);
    if (nk > 0) {
        load_tile_fast<0, BM>(p.A, p.lda, kbeg, offa, ra);
        load_tile_fast<BLAY, BN>(p.B, p.ldb, kbeg, offb, rb);
        store_tile_rk<BM>(As, tid, ra);
        if (BLAY == 0) store_tile_rk<BN>(Bs, tid, rb);
        else store_tile<1, BN>(Bs, tid, rb);
    }
    __syncthreads();

    // per-lane fragment addresses inside a buffer (floats): row * BK + ((chunk ^ swz(row)) << 2), chunk = 2 * half + c
    int a_off[MT][2], b_off[NT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wm * WM + i * 32 + l31;
#pragma unroll
        for (int c = 0; c < 2; ++c) a_off[i][c] = row * BK + (((2 * half + c) ^ ((row >> 2) & 3)) << 2);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 32 + l31;
#pragma unroll
        for (int c = 0; c < 2; ++c) b_off[j][c] = row * BK + (((2 * half + c) ^ ((row >> 2) & 3)) << 2);
    }

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int kn = kbeg + min(kt + 1, nk - 1) * BK;
        load_tile_fast<0, BM>(p.A, p.lda, kn, offa, ra);
        load_tile_fast<BLAY, BN>(p.B, p.ldb, kn, offb, rb);
        const float* Ac = As + cur * BK * BM;
        const float* Bc = Bs + cur * BK * BN;
        f32x4 av[MT][2], bv[NT][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i][c] = *reinterpret_cast<const f32x4*>(Ac + a_off[i][c]);
            if (BLAY == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j][c] = *reinterpret_cast<const f32x4*>(Bc + b_off[j][c]);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float bs[NT];
            if (BLAY == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bs[j] = bv[j][s >> 2][s & 3];
            } else {      // [k][row] image, k = 8 * half + s
                const int k = 8 * half + s;
                const int sw = (((k >> 2) & 3) << 3) ^ ((k & 1) << 4);
#pragma unroll
                for (int j = 0; j < NT; ++j) bs[j] = Bc[k * BN + ((wn * WN + j * 32 + l31) ^ sw)];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(av[i][s >> 2][s & 3], bs[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
            store_tile_rk<BM>(As + (cur ^ 1) * BK * BM, tid, ra);
            if (BLAY == 0) store_tile_rk<BN>(Bs + (cur ^ 1) * BK * BN, tid, rb);
            else store_tile<1, BN>(Bs + (cur ^ 1) * BK * BN, tid, rb);
        }
        __syncthreads();
    }

    float* Cb = p.C + (long)blockIdx.z * p.slab_stride;
    const int epi = (gridDim.z > 1) ? (int)EPI_NONE : p.epi;
    store_tile_dispatch_lds<MT, NT>(epi, p, Cb, acc, row0 + wm * WM, col0 + wn * WN, lane, smem + wave * epi_stage_floats(NT));
}

// Deterministic split-K combine: C (+)= sum over slabs in slab order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                            long ldc, int M, int N, int slabs, long slab_stride,
                                                            int accumulate, const float* __restrict__ colsum_ws,
                                                            float* __restrict__ colsum_out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)M * N;
    if (idx >= total) {   // tail threads fold the fused bias-gradient slabs
        const long m = idx - total;
        if (colsum_out && m < M) {
            float s = 0.f;
            for (int z = 0; z < slabs; ++z) s += colsum_ws[(long)z * M + m];
            colsum_out[m] = accumulate ? colsum_out[m] + s : s;
        }
        return;
    }
    const int m = (int)(idx / N), n = (int)(idx % N);
    float s = 0.f;
    for (int z = 0; z < slabs; ++z) s += ws[(long)z * slab_stride + (long)m * N + n];
    float* cp = C + (long)m * ldc + n;
    *cp = accumulate ? (*cp + s) : s;
}

// 16-byte form of the same combine (N, ldc multiples of 4, 16-byte aligned bases): one thread per 4 consecutive columns, all slab loads
// of a thread independent and in flight together.  The bias-gradient slabs are folded by the tail threads as above.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ ws, float* __restrict__ C, long ldc, int M, int N,
                                                                int slabs, long slab_stride, int accumulate,
                                                                const float* __restrict__ colsum_ws, float* __restrict__ colsum_out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nq = N >> 2;
    const long total = (long)M * nq;
    if (idx >= total) {
        const long m = idx - total;
        if (colsum_out && m < M) {
            float s = 0.f;
            for (int z = 0; z < slabs; ++z) s += colsum_ws[(long)z * M + m];
            colsum_out[m] = accumulate ? colsum_out[m] + s : s;
        }
        return;
    }
    const int m = (int)(idx / nq), n = (int)(idx % nq) * 4;
    const float* src = ws + (long)m * N + n;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < slabs; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)z * slab_stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += v[j];
    }
    f32x4* cp = reinterpret_cast<f32x4*>(C + (long)m * ldc + n);
    if (accumulate) {
        const f32x4 o = *cp;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += o[j];
    }
    *cp = s;
}

// Split-K combine for the forward / dgrad layouts: sums the slabs in slab order and applies the epilogue the single-pass kernel
// would have applied in its store (bias, ReLU / GELU (+ pre-activation copy), residual, activation-gradient masks, positional row add
// with the CLS row remap, accumulate).  `p` carries the REAL output (C, ldc) and epilogue operands.
__global__ __launch_bounds__(256) void splitk_reduce_epi_kernel(GemmArgs p, const float* __restrict__ ws, int slabs, long slab_stride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * p.N) return;
    const int m = (int)(idx / p.N), n = (int)(idx % p.N);
    float v = 0.f;
    for (int z = 0; z < slabs; ++z) v += ws[(long)z * slab_stride + idx];
    const int e = p.epi;
    const bool has_bias = e == EPI_BIAS || e == EPI_BIAS_RELU || e == EPI_BIAS_GELU || e == EPI_BIAS_RESID || e == EPI_BIAS_ROWADD;
    if (has_bias && p.bias) v += p.bias[n];
    long orow = m;
    if (e == EPI_BIAS_ROWADD && p.remap_rows_per_img > 0) orow = (long)m + (long)(m / p.remap_rows_per_img) * p.remap_skip + p.remap_skip;
    float* cp = p.C + orow * p.ldc + n;
    if (e == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
    if (e == EPI_BIAS_GELU) {
        if (p.aux_out) p.aux_out[(long)m * p.ldc + n] = v;
        v = gelu_erf(v);
    }
    if (e == EPI_BIAS_RESID) v += p.aux[(long)m * p.ldaux + n];
    if (e == EPI_MUL_GELU_GRAD) v *= gelu_erf_grad(p.aux[(long)m * p.ldaux + n]);
    if (e == EPI_MUL_RELU_MASK) v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
    if (e == EPI_BIAS_ROWADD) v += p.aux[(long)(m % p.aux_rows) * p.ldaux + n];
    if (e == EPI_ACCUM) v += *cp;
    *cp = v;
}

// dst[c][r] = src[r][c] for a [R][C] matrix (weights): 64x64 tiles through LDS, 16-byte accesses on both sides.  Used to keep a
// k-contiguous copy W^T of a Linear weight so that the input-gradient GEMM dx = dy W runs in the NT layout (both operands k-contiguous:
// the [row][k] LDS image with 16-byte LDS traffic) instead of NN.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, r = f >> 4, c4 = (f & 15) * 4;
        if (r0 + r < R) {
            if (c0 + c4 + 3 < C && ((C & 3) == 0)) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)(r0 + r) * C + c0 + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[r][c4 + j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + c4 + j < C) t[r][c4 + j] = src[(long)(r0 + r) * C + c0 + c4 + j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, c = f >> 4, r4 = (f & 15) * 4;
        if (c0 + c < C) {
            if (r0 + r4 + 3 < R && ((R & 3) == 0)) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = t[r4 + j][c];
                *reinterpret_cast<f32x4*>(dst + (long)(c0 + c) * R + r0 + r4) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + r4 + j < R) dst[(long)(c0 + c) * R + r0 + r4 + j] = t[r4 + j][c];
            }
        }
    }
}

// Batched form: one launch transposes every weight matrix of a parameter arena into a second arena.  desc[block] = {source offset,
// destination offset (floats, relative to the two bases), R, C, r0, c0}: the block moves the 64x64 tile at (r0, c0) of that matrix.
struct TransposeTile { long src_off, dst_off; int R, C, r0, c0; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                                const TransposeTile* __restrict__ desc) {
    __shared__ float t[64][65];
    const TransposeTile d = desc[blockIdx.x];
    const float* src = src_base + d.src_off;
    float* dst = dst_base + d.dst_off;
    const int R = d.R, C = d.C, r0 = d.r0, c0 = d.c0, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, r = f >> 4, c4 = (f & 15) * 4;
        if (r0 + r < R) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + c4 + j < C) t[r][c4 + j] = src[(long)(r0 + r) * C + c0 + c4 + j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, c = f >> 4, r4 = (f & 15) * 4;
        if (c0 + c < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + r4 + j < R) dst[(long)(c0 + c) * R + r0 + r4 + j] = t[r4 + j][c];
        }
    }
}

// Column sums (bias gradients): out[n] (+)= sum_m X[m][n].  Stage 1: each block owns 64 columns and a row slice.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long ldx, int M, int N,
                                                             int rows_per_slice, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
    float s = 0.f;
    if (n < N)
        for (int m = m0 + wave; m < m1; m += 4) s += X[(long)m * ldx + n];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
// Fold of MANY partial rows (hundreds): 1024 threads = 16 columns x 64 partial-lanes, each thread adds every 64th partial of its
// column (independent loads), LDS combines the 64 lane sums of a column in a fixed order.  grid = ceil(N / 16).
__global__ __launch_bounds__(1024) void colsum_fold_wide_kernel(const float* __restrict__ part, int N, int slices, float* __restrict__ out,
                                                                int accumulate) {
    __shared__ float red[64][17];
    const int col = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + col;
    float s = 0.f;
    if (n < N)
        for (int z = kl; z < slices; z += 64) s += part[(long)z * N + n];
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && n < N) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) t += red[k][col];
        out[n] = accumulate ? out[n] + t : t;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int N, int slices,
                                                           float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int z = 0; z < slices; ++z) s += part[(long)z * N + n];
    out[n] = accumulate ? out[n] + s : s;
}


// Tile choice.  All workgroups of one launch cost the same, and a CU's matrix pipes are the bottleneck, so the launch
// lasts ceil(nWG / 256 CUs) "rounds" of one tile's work: pick the tile that minimises rounds * tile area (small
// penalty for the smaller tiles' extra LDS traffic per MFMA).  N = 384 GEMMs at B*n = 12672 rows go from 58 % CU
// fill with 128x128 tiles (297 workgroups) to 93 % with 64x64 (1188).
struct Tile { int bm, bn; float penalty; };
inline Tile pick_tile(int M, int N) {
    static const int forced = [] { const char* e = getenv("D2S_GEMM_TILE"); return e ? atoi(e) : 0; }();   // tuning aid: 1..4 = candidate index
    // per-tile cost factors, re-measured on the [row][k]-image kernel (D2S_GEMM_TILE sweep, profiles/r02_d_gemm_tile_sweep.txt; SQ counters:
    // the 64x64 tile spends 1.9 vector + 2.1 scalar instructions per MFMA on addressing / loop control, the 128x64 tile 1.1 + 1.1)
    const Tile cand[4] = {{128, 128, 1.00f}, {128, 64, 1.04f}, {64, 128, 1.05f}, {64, 64, 1.10f}};
    if (forced >= 1 && forced <= 4) return cand[forced - 1];
    Tile best = cand[0];
    float best_cost = 1e30f;
    for (const Tile& t : cand) {
        const long nwg = (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
        const long per_cu = (nwg + 255) / 256;                      // workgroups the busiest CU gets
        // one workgroup puts one wave on each SIMD; a single wave cannot keep the matrix pipe busy (dependent MFMAs on its few
        // accumulators, exposed LDS / barrier latency), two nearly can: small grids should prefer more, smaller workgroups
        const float pipe = per_cu >= 3 ? 1.0f : per_cu == 2 ? 0.85f : 0.6f;
        const float cost = (float)per_cu * t.bm * t.bn * t.penalty / pipe;
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

// Residency cap: a bare loop of f32 MFMAs issues at 0.99 of the pipe's rate with one or two waves per SIMD and at 0.80 with four
// (tools/micro/mfma_shape_f32.hip, profiles/r02_mfma_residency.txt), so more co-resident workgroups than the latency hiding needs cost
// matrix throughput.  Unused dynamic LDS is the knob: asking for 160 KB / k - static bytes per workgroup lets a CU hold exactly k.
inline unsigned residency_pad(int bm, int bn) {
    static const int cap = [] { const char* e = getenv("D2S_GEMM_WG_PER_CU"); return e ? atoi(e) : 0; }();
    if (cap <= 0) return 0;
    const int static_bytes = 2 * BK * (bm + bn) * (int)sizeof(float) + 64;
    const int want = (160 * 1024) / cap - static_bytes - 512;
    return want > 0 ? (unsigned)(want & ~255) : 0u;
}
template <int ALAY, int BLAY, bool FAST>
inline void launch_gemm_f(const Tile& t, dim3 grid, hipStream_t stream, const GemmArgs& p) {
    dim3 block(256);
    const unsigned pad = residency_pad(t.bm, t.bn);
    if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((gemm_f32_kernel<ALAY, BLAY, 128, 128, FAST>), grid, block, pad, stream, p);
    else if (t.bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<ALAY, BLAY, 128, 64, FAST>), grid, block, pad, stream, p);
    else if (t.bn == 128) hipLaunchKernelGGL((gemm_f32_kernel<ALAY, BLAY, 64, 128, FAST>), grid, block, pad, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<ALAY, BLAY, 64, 64, FAST>), grid, block, pad, stream, p);
}
template <int ALAY, int BLAY>
inline void launch_gemm16(const Tile& t, dim3 grid, hipStream_t stream, const GemmArgs& p) {
    dim3 block(256);
    const unsigned pad = residency_pad(t.bm, t.bn);
    if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((gemm_f32_kernel16<ALAY, BLAY, 128, 128>), grid, block, pad, stream, p);
    else if (t.bm == 128) hipLaunchKernelGGL((gemm_f32_kernel16<ALAY, BLAY, 128, 64>), grid, block, pad, stream, p);
    else if (t.bn == 128) hipLaunchKernelGGL((gemm_f32_kernel16<ALAY, BLAY, 64, 128>), grid, block, pad, stream, p);
    else hipLaunchKernelGGL((gemm_f32_kernel16<ALAY, BLAY, 64, 64>), grid, block, pad, stream, p);
}
template <int BLAY>
inline void launch_gemm_rk(const Tile& t, dim3 grid, hipStream_t stream, const GemmArgs& p) {
    dim3 block(256);
    const unsigned pad = residency_pad(t.bm, t.bn);
    if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((gemm_f32_rk_kernel<BLAY, 128, 128>), grid, block, pad, stream, p);
    else if (t.bm == 128) hipLaunchKernelGGL((gemm_f32_rk_kernel<BLAY, 128, 64>), grid, block, pad, stream, p);
    else if (t.bn == 128) hipLaunchKernelGGL((gemm_f32_rk_kernel<BLAY, 64, 128>), grid, block, pad, stream, p);
    else hipLaunchKernelGGL((gemm_f32_rk_kernel<BLAY, 64, 64>), grid, block, pad, stream, p);
}
// D2S_GEMM_RK [1]: k-contiguous A operands (NT / NN layouts, guard-free shapes) use the [row][k] LDS image (ds_*_b128); 0 = the [k][row]
// image everywhere.  Measured on the model's shapes at B = 128 (tools/gemm_bench.py, profiles/r02_c_gemm_rk_vs_krow.txt): +20...30 % on
// the N = 384 outputs and the K = 1536 / 1152 reductions, +4...7 % on the wide outputs, within +-2 % on the n = 99 input-gradient shapes
inline bool use_rk() {
    static const int on = [] { const char* e = getenv("D2S_GEMM_RK"); return e ? atoi(e) : 1; }();
    return on != 0;
}
// D2S_GEMM_MFMA16: 1 = run the guard-free shapes on the 16x16x4 form, 0 = always the 32x32x2 form
inline bool use_mfma16() {
    static const int on = [] { const char* e = getenv("D2S_GEMM_MFMA16"); return e ? atoi(e) : 0; }();
    return on != 0;
}
template <int ALAY, int BLAY>
inline void launch_gemm(const Tile& t, dim3 grid, hipStream_t stream, const GemmArgs& p, bool fast) {
    if (fast && p.vec_epilogue && ALAY == 0 && use_rk()) launch_gemm_rk<BLAY>(t, grid, stream, p);
    else if (fast && p.vec_epilogue && use_mfma16()) launch_gemm16<ALAY, BLAY>(t, grid, stream, p);
    else if (fast) launch_gemm_f<ALAY, BLAY, true>(t, grid, stream, p);
    else launch_gemm_f<ALAY, BLAY, false>(t, grid, stream, p);
}

}  // namespace

namespace d2s_gemm {
size_t split_workspace_bytes(int split, int M, int N, int K);
int launch_split_gemm(const GemmArgs& p, int b_cols, int split, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t split_tn_pieces_bytes(int split, int M, int N, int K);
int split_tn_colsum_partials(int K);
bool split_tn_use_dma(int M, int N, int K);
int split_tn_dma_slices(int M, int N, int K);
int launch_split_gemm_tn(const GemmArgs& p, int split, int slices, void* pieces_ws, float* colsum_part, hipStream_t stream, int* colsum_parts);
}
// Workgroups the wgrad launch aims for (tiles x K slices, rounded down).  Run alone, exactly 2 per CU is best (tools/gemm_bench.py,
// D2S_SPLITK_TARGET sweep in round 1: 512 beats 768 / 1024 by 4-15 % - fewer, longer K slices mean less slab traffic for the ordered
// combine - and any count that is not a multiple of the CU count loses 10-40 % to imbalance).  Inside the training step the weight
// gradients run on their own stream BESIDE the dgrad / attention kernels (d2s.ops.async_weight_grads), where shorter workgroups interleave
// better: 3 per CU measured best there (step: 256 -> 3483, 384 -> 3470, 512 -> 3513, 768 -> 3556, 1024 -> 3526, 1536 -> 3508, 2048 -> 3490
// images/s on one box), so that is the default.
static int splitk_target() {
    static const int t = [] {
        const char* e = getenv("D2S_SPLITK_TARGET");
        if (e) return atoi(e);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return 3 * cus;
    }();
    return t;
}
static inline size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }
// mode 2 only: run the weight gradient on the bf16 matrix cores too (D2S_BF16_WGRAD=0 keeps it on the exact fp32 kernel)
static bool bf16_wgrad(int mode) {
    static const int on = [] { const char* e = getenv("D2S_BF16_WGRAD"); return e ? atoi(e) : 1; }();
    return mode == 2 && on;
}
extern "C" size_t d2s_colsum_workspace_bytes(int M, int N);
extern "C" int d2s_colsum_f32(const float* X, long ldx, int M, int N, float* out, int accumulate, void* workspace,
                              size_t workspace_bytes, hipStream_t stream);
// Forward / dgrad GEMMs whose tile grid leaves most CUs with 0-2 workgroups (small batches, narrow outputs with a long K) also
// split K: the choice minimises (workgroups on the busiest CU) x (K share per workgroup) / (matrix-pipe use at that residency) plus a
// per-slice cost for the extra slab traffic.  Large grids get 1 (no split).
static int nt_slices(int tiles, int K) {
    static const int enabled = [] { const char* e = getenv("D2S_NT_SPLITK"); return e ? atoi(e) : 1; }();
    // only grids that leave the busiest CU with at most 2 workgroups: with more, dispatch evens the load out by itself and the split
    // only adds slab traffic (measured at B=128: -4...-19 % on the shapes an unrestricted model chose to split)
    if (!enabled || K < 512 || tiles > 512) return 1;
    int best = 1;
    float best_cost = 1e30f;
    const int smax = K / 256 < 8 ? K / 256 : 8;
    for (int sl = 1; sl <= smax; ++sl) {
        const long per_cu = ((long)tiles * sl + 255) / 256;
        const float pipe = per_cu >= 3 ? 1.0f : per_cu == 2 ? 0.85f : 0.6f;
        const float cost = (float)per_cu / ((float)sl * pipe) + 0.04f * (sl - 1);
        if (cost < best_cost - 1e-6f) { best_cost = cost; best = sl; }
    }
    return best;
}
static int splitk_slices(int tiles, int K) {
    const int target = splitk_target();
    int slices = target >= 100000 ? (target - 100000 + tiles - 1) / tiles : target / tiles;   // >= 100000: round up, else round down
    const int max_slices = (K + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    return slices;
}

#ifdef D2S_STAMPS
extern "C" int d2s_debug_read_stamps_f32(unsigned long long* host_out, int n_wg) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps_f32), (size_t)n_wg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

// Workspace needed by d2s_gemm_f32 for a given problem and arithmetic mode (pass the same mode to the query and to the call).
// mode 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  1: bf16x3 split on the bf16 matrix cores, fp32-class accuracy (gemm_split.hip).
// 2: bf16 operands, fp32 accumulation.  Modes 1 and 2 apply to the NT and NN layouts; in mode 2 wgrad (TN) also runs on the bf16
// matrix cores (transposing split + K-sliced pieces kernel + the same ordered slab combine), in mode 1 it stays on the exact kernel.
// The mode is an argument of every call (no process-global state): a bf16 teacher can run beside an fp32 student, from any thread.
size_t d2s_gemm_f32_workspace_bytes(int layout, int M, int N, int K, int mode) {
    if (layout != 2 && mode != 0) return split_workspace_bytes(mode == 1 ? 3 : 1, M, N, K);   // bf16 piece matrices
    if (layout != 2) {
        const Tile t = pick_tile(M, N);
        const int sl = nt_slices(((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn), K);
        return sl > 1 ? (size_t)sl * M * N * sizeof(float) : 0;
    }
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);      // (64x64 wgrad tiles for the small 384x384 weights: measured slower)
    int slices = splitk_slices(tiles, K);
    size_t bytes = slices <= 1 ? 0 : ((size_t)slices * M * N + (size_t)slices * M) * sizeof(float);   // C slabs + fused bias-gradient slabs
    if (bf16_wgrad(mode) && split_tn_use_dma(M, N, K)) slices = split_tn_dma_slices(M, N, K);
    if (bf16_wgrad(mode))   // + bf16 pieces of both operands + scratch of the separate bias-gradient pass (see gemm_impl)
        bytes = ws_align(((size_t)(slices + 1) * M * N) * sizeof(float)) + ws_align(split_tn_pieces_bytes(1, M, N, K)) + (size_t)split_tn_colsum_partials(K) * M * sizeof(float);
    return bytes;
}

// layout 0 = NT (A[M,K], B[N,K]); 1 = NN (A[M,K], B[K,N]); 2 = TN (A[K,M], B[K,N]).
// epilogue: see enum Epi.  accumulate != 0 (TN only): C += result.  remap_*: see GemmArgs.
static int gemm_impl(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                     int K, int epilogue, const float* bias, const float* aux, long ldaux, float* aux_out, int aux_rows,
                     int remap_rows_per_img, int remap_skip, int accumulate, void* workspace, size_t workspace_bytes,
                     hipStream_t stream, float* colsum_out, int mode, const void* a16 = nullptr, void* c16 = nullptr, const void* b16 = nullptr) {
    // layout 2 (weight gradient): a16 is the bf16 form of the SECOND operand (the layer input x); B may then be NULL
    // layout 2: b16 is the bf16 form of the FIRST operand (dy); A may then be NULL
    if (b16 && (mode != 2 || (layout != 2 && (K % 32 != 0 || !aligned16(b16))) || (layout == 2 && (!bf16_wgrad(mode) || (reinterpret_cast<uintptr_t>(b16) & 7))))) return D2S_ERR_ARG;
    if ((!A && !(a16 && layout != 2) && !(b16 && layout == 2)) || (!B && !(a16 && layout == 2) && !(b16 && layout != 2)) || (!C && !c16) || M <= 0 || N <= 0 || K <= 0 || layout < 0 || layout > 2 || mode < 0 || mode > 2) return D2S_ERR_ARG;
    if ((a16 || c16) && (mode != 2 || (layout != 2 && accumulate) || remap_rows_per_img > 0)) return D2S_ERR_ARG;   // bf16 operands / copies exist in bf16 mode only
    if (layout == 2 && (c16 || (a16 && !bf16_wgrad(mode)))) return D2S_ERR_ARG;
    if (layout == 2 && a16) {
        if ((reinterpret_cast<uintptr_t>(a16) & 7) != 0) return D2S_ERR_ARG;
    }
    if ((a16 && layout != 2 && (K % 32 != 0 || !aligned16(a16))) || (c16 && (N % 32 != 0 || !aligned16(c16)))) return D2S_ERR_ARG;
    int aux_bf16 = 0;       // bf16 data path: the GELU pre-activation saved / read in bf16 (d2s_gemm_f32_bf16io, codes 9 / 10)
    if (epilogue == EPI_BIAS_GELU_Z16 || epilogue == EPI_MUL_GELU_GRAD_Z16) {
        const float* z = epilogue == EPI_BIAS_GELU_Z16 ? aux_out : aux;
        if (mode != 2 || layout == 2 || !z || (reinterpret_cast<uintptr_t>(z) & 7) || N % 4 != 0 || (epilogue == EPI_MUL_GELU_GRAD_Z16 && ldaux % 4 != 0) ||
            (epilogue == EPI_BIAS_GELU_Z16 && ldc % 4 != 0))
            return D2S_ERR_ARG;
        aux_bf16 = 1;
        epilogue = epilogue == EPI_BIAS_GELU_Z16 ? EPI_BIAS_GELU : EPI_MUL_GELU_GRAD;
    }
    if (epilogue < EPI_NONE || epilogue > EPI_ACCUM) return D2S_ERR_ARG;
    if ((epilogue == EPI_BIAS_RESID || epilogue == EPI_MUL_GELU_GRAD || epilogue == EPI_MUL_RELU_MASK ||
         epilogue == EPI_BIAS_ROWADD) && !aux)
        return D2S_ERR_ARG;
    GemmArgs p;
    p.aux_bf16 = aux_bf16;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux = aux; p.aux_out = aux_out;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.aux_rows = aux_rows > 0 ? aux_rows : 1;
    p.remap_rows_per_img = remap_rows_per_img; p.remap_skip = remap_skip;
    p.colsum = nullptr; p.colsum_accumulate = accumulate;
    p.a16 = a16; p.c16 = c16; p.b16 = b16;
    {
        p.stagger = 0;
        p.vec_epilogue = 0;
    }
    const int alay = layout == 2 ? 1 : 0, blay = layout == 0 ? 0 : 1;
    p.vecA = aligned16(A) && (lda % 4 == 0) && (alay == 0 ? (K % 4 == 0) : (M % 4 == 0));
    p.vecB = aligned16(B) && (ldb % 4 == 0) && (blay == 0 ? (K % 4 == 0) : (N % 4 == 0));
    if (mode != 0 && layout != 2) {
        if (accumulate) p.epi = EPI_ACCUM;
        p.k_per_slice = K;
        p.slab_stride = 0;
        return launch_split_gemm(p, layout == 1 ? 1 : 0, mode == 1 ? 3 : 1, workspace, workspace_bytes, stream);
    }
    if (layout == 2 && bf16_wgrad(mode)) {
        // wgrad in bf16 mode: transposing split of dy and x into K-contiguous bf16 pieces, K-sliced pieces kernel into fp32 slabs,
        // the same ordered slab combine as the exact path; the bias gradient is a separate exact fp32 column-sum pass over dy.
        const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        int slices = split_tn_use_dma(M, N, K) ? split_tn_dma_slices(M, N, K) : splitk_slices(tiles, K);
        const int Kp = ((K + 63) / 64) * 64;
        int kper = (((Kp + slices - 1) / slices + 63) / 64) * 64;
        slices = (Kp + kper - 1) / kper;      // never more than the query assumed
        const size_t slab_bytes = ws_align(((size_t)(slices + 1) * M * N) * sizeof(float));
        const size_t piece_bytes = ws_align(split_tn_pieces_bytes(1, M, N, K));
        const size_t cs_bytes = (size_t)split_tn_colsum_partials(K) * M * sizeof(float);
        if (!workspace || workspace_bytes < slab_bytes + piece_bytes + cs_bytes) return D2S_ERR_WORKSPACE;
        unsigned char* wsb = static_cast<unsigned char*>(workspace);
        p.k_per_slice = kper;
        p.epi = EPI_NONE;
        p.C = reinterpret_cast<float*>(wsb);
        p.ldc = N;
        p.slab_stride = (long)M * N;
        p.remap_rows_per_img = 0;
        float* cs_part = colsum_out ? reinterpret_cast<float*>(wsb + slab_bytes + piece_bytes) : nullptr;
        int cs_parts = 0;
        int rc = launch_split_gemm_tn(p, 1, slices, wsb + slab_bytes, cs_part, stream, &cs_parts);
        if (rc != D2S_OK) return rc;
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const float*>(wsb), C, ldc, M, N, slices, (long)M * N, accumulate,
                           static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
        if (colsum_out)   // ordered fold of the per-64-token partials that the split pass of dy produced (exact fp32)
            hipLaunchKernelGGL(colsum_fold_wide_kernel, dim3((M + 15) / 16), dim3(1024), 0, stream, cs_part, M, cs_parts,
                               colsum_out, accumulate);
        return d2s_check_launch();
    }
    Tile tile = layout == 2 ? Tile{128, 128, 1.f} : pick_tile(M, N);
    const int tiles = ((M + tile.bm - 1) / tile.bm) * ((N + tile.bn - 1) / tile.bn);
    int slices = layout == 2 ? splitk_slices(tiles, K) : nt_slices(tiles, K);
    int kper = (K + slices - 1) / slices;
    kper = ((kper + BK - 1) / BK) * BK;
    slices = (K + kper - 1) / kper;
    p.k_per_slice = kper;
    p.slab_stride = 0;
    float* realC = C;
    if (accumulate && layout != 2 && p.epi == EPI_NONE) p.epi = EPI_ACCUM;
    const GemmArgs real = p;              // output and epilogue operands as the caller gave them (used by the NT / NN slab combine)
    if (slices > 1 && layout != 2) {
        const size_t need = (size_t)slices * M * N * sizeof(float);
        if (!workspace || workspace_bytes < need) return D2S_ERR_WORKSPACE;
        p.C = static_cast<float*>(workspace);
        p.ldc = N;
        p.slab_stride = (long)M * N;
        p.remap_rows_per_img = 0;
    } else if (slices > 1) {
        const size_t need = ((size_t)slices * M * N + (size_t)slices * M) * sizeof(float);
        if (!workspace || workspace_bytes < need) return D2S_ERR_WORKSPACE;
        p.C = static_cast<float*>(workspace);
        p.ldc = N;
        p.slab_stride = (long)M * N;
        p.remap_rows_per_img = 0;
        if (colsum_out) p.colsum = static_cast<float*>(workspace) + (size_t)slices * M * N;
    } else {
        if (accumulate) p.epi = EPI_ACCUM;
        p.colsum = colsum_out;
    }
    {
        static const int vec_epi_env = [] { const char* e = getenv("D2S_GEMM_VEC_EPILOGUE"); return e ? atoi(e) : 1; }();
        p.vec_epilogue = (vec_epi_env && epilogue_vec_ok(p) && (p.slab_stride % 4 == 0)) ? 1 : 0;
    }
    dim3 grid(tiles, 1, slices), block(256);
    // guard-free instantiation: both operands 16-byte loadable, every K slice a whole number of K-steps, and at least one full
    // 4-row chunk where rows are the contiguous dimension (the clamp needs rows - 4 >= 0)
    static const int fast_env = [] { const char* e = getenv("D2S_GEMM_FAST"); return e ? atoi(e) : 1; }();
    const bool fast = fast_env && p.vecA && p.vecB && (K % BK == 0) && (alay == 0 || M >= 4) && (blay == 0 || N >= 4);
    if (layout == 0) launch_gemm<0, 0>(tile, grid, stream, p, fast);
    else if (layout == 1) launch_gemm<0, 1>(tile, grid, stream, p, fast);
    else launch_gemm<1, 1>(tile, grid, stream, p, fast);
    if (slices > 1 && layout != 2) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_epi_kernel, dim3((unsigned)((total + 255) / 256)), block, 0, stream, real,
                           static_cast<const float*>(workspace), slices, (long)M * N);
    } else if (slices > 1) {
        if ((N % 4 == 0) && (ldc % 4 == 0) && aligned16(realC) && aligned16(workspace) && (((long)M * N) % 4 == 0)) {
            const long total = (long)M * (N / 4) + (colsum_out ? M : 0);
            hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)((total + 255) / 256)), block, 0, stream,
                               static_cast<const float*>(workspace), realC, ldc, M, N, slices, (long)M * N, accumulate, p.colsum, colsum_out);
        } else {
            const long total = (long)M * N + (colsum_out ? M : 0);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), block, 0, stream,
                               static_cast<const float*>(workspace), realC, ldc, M, N, slices, (long)M * N, accumulate,
                               p.colsum, colsum_out);
        }
    }
    return d2s_check_launch();
}

int d2s_gemm_f32(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                 int K, int epilogue, const float* bias, const float* aux, long ldaux, float* aux_out, int aux_rows,
                 int remap_rows_per_img, int remap_skip, int accumulate, int mode, void* workspace, size_t workspace_bytes,
                 hipStream_t stream) {
    return gemm_impl(layout, A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, aux, ldaux, aux_out, aux_rows, remap_rows_per_img,
                     remap_skip, accumulate, workspace, workspace_bytes, stream, nullptr, mode);
}

// d2s_gemm_f32 in mode 2 (bf16 operands) with bf16 side channels: a_bf16 (optional) = the A operand already rounded to bf16, dense
// [M][K] - the call then skips its conversion pass over A, which must still hold the same values in fp32; c_bf16 (optional) = a dense
// [M][N] bf16 copy of the result written by the same epilogue - the a_bf16 of the next GEMM.  NT / NN layouts, K % 32 == 0 for a_bf16,
// N % 32 == 0 for c_bf16.  With a_bf16 given A may be NULL, with c_bf16 given C may be NULL (forward-only passes that keep no fp32 form).
// b_bf16 (optional) = the B operand in bf16, ALWAYS [N][K] k-contiguous (for layout 1, dx = dy W, that is W^T): a cached weight
// (d2s_convert_bf16 once per optimiser step, or once for a frozen model) - the call then skips its per-call conversion of B; B may be NULL.
int d2s_gemm_f32_bf16io(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                        int epilogue, const float* bias, const float* aux, long ldaux, float* aux_out, const void* a_bf16, const void* b_bf16,
                        void* c_bf16, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (layout == 2) return D2S_ERR_ARG;      // weight gradients with a bf16 input: d2s_linear_wgrad_f32_bf16x
    return gemm_impl(layout, A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, aux, ldaux, aux_out, 0, 0, 0, 0, workspace, workspace_bytes,
                     stream, nullptr, 2, a_bf16, c_bf16, b_bf16);
}

// nn.Linear backward w.r.t. its parameters in one pass over dy:  dW[n_out, n_in] (+)= dy^T x,  db[n_out] (+)= column sums of dy.
// The bias gradient is folded out of the dy tiles the GEMM streams through its registers anyway (no second read of dy).
size_t d2s_linear_wgrad_workspace_bytes(int tokens, int n_out, int n_in, int mode) {
    return d2s_gemm_f32_workspace_bytes(2, n_out, n_in, tokens, mode);
}
int d2s_linear_wgrad_f32(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, float* db, int tokens,
                         int n_out, int n_in, int accumulate, int mode, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return gemm_impl(2, dy, lddy, x, ldx, dW, lddw, n_out, n_in, tokens, EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0,
                     accumulate, workspace, workspace_bytes, stream, db, mode);
}
// The same in mode 2 with the layer input given in bf16 only (x_bf16 [tokens][n_in], row stride ldx elements): what the bf16 data path
// saved for the backward instead of the fp32 activation.  dy is fp32 (its exact column sums are the bias gradient) or, when the kernel that
// produced it wrote bf16 only, dy_bf16 [tokens][n_out] (dy may be NULL; the bias gradient is then the sum of the bf16 values).  BOTH may be
// given (dy_bf16 = dy rounded to bf16, e.g. by the LayerNorm backward that produced dy): the matrix kernel then reads the two bf16 operands
// token-major as they lie - no transposing pass - and the bias gradient is still the exact column sum of dy; results are bit-identical
// to the dy-only call.
int d2s_linear_wgrad_f32_bf16x(const float* dy, const void* dy_bf16, long lddy, const void* x_bf16, long ldx, float* dW, long lddw, float* db,
                               int tokens, int n_out, int n_in, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x_bf16 || (!dy && !dy_bf16)) return D2S_ERR_ARG;
    return gemm_impl(2, dy, lddy, nullptr, ldx, dW, lddw, n_out, n_in, tokens, EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0,
                     accumulate, workspace, workspace_bytes, stream, db, 2, x_bf16, nullptr, dy_bf16);
}

// dst[C][R] = src[R][C]^T (dense row-major): the k-contiguous copy of a Linear weight for the input-gradient GEMM.
int d2s_transpose_f32(const float* src, float* dst, int R, int C, hipStream_t stream) {
    if (!src || !dst || R <= 0 || C <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, src, dst, R, C);
    return d2s_check_launch();
}

// n_tiles descriptors of 6 fields {long src_off, long dst_off, int R, int C, int r0, int c0} (32 bytes each) in device memory.
int d2s_transpose_batched_f32(const float* src_base, float* dst_base, const void* tile_desc, int n_tiles, hipStream_t stream) {
    if (!src_base || !dst_base || !tile_desc || n_tiles <= 0) return D2S_ERR_ARG;
    static_assert(sizeof(TransposeTile) == 32, "descriptor layout is part of the C ABI");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(n_tiles), dim3(256), 0, stream, src_base, dst_base, static_cast<const TransposeTile*>(tile_desc));
    return d2s_check_launch();
}

size_t d2s_colsum_workspace_bytes(int M, int N) {
    int slices = (M + 511) / 512;
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    return (size_t)slices * N * sizeof(float);
}

// out[n] (+)= sum_m X[m][n]  - bias gradients of every Linear on the path.
int d2s_colsum_f32(const float* X, long ldx, int M, int N, float* out, int accumulate, void* workspace,
                   size_t workspace_bytes, hipStream_t stream) {
    if (!X || !out || M <= 0 || N <= 0) return D2S_ERR_ARG;
    int slices = (M + 511) / 512;
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    const int rps = (M + slices - 1) / slices;
    slices = (M + rps - 1) / rps;
    if (!workspace || workspace_bytes < (size_t)slices * N * sizeof(float)) return D2S_ERR_WORKSPACE;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, slices), dim3(256), 0, stream, X, ldx, M, N, rps, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, slices, out, accumulate);
    return d2s_check_launch();
}

}  // extern "C"
