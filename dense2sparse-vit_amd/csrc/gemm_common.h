// Shared pieces of the GEMM kernels (gemm_f32.hip: exact f32 MFMA; gemm_split.hip: bf16-split MFMA): argument block,
// epilogue kinds and the accumulator -> memory epilogue for 32x32 MFMA accumulator tiles (C/D layout is dtype independent).
#pragma once
#include "d2s_common.h"

namespace d2s_gemm {

enum Epi : int {
    EPI_NONE = 0,
    EPI_BIAS = 1,          // C = acc + bias[n]
    EPI_BIAS_RELU = 2,     // C = relu(acc + bias[n])
    EPI_BIAS_GELU = 3,     // aux_out = acc + bias[n] (pre-activation, if given); C = gelu(.)
    EPI_BIAS_RESID = 4,    // C = acc + bias[n] + aux[m][n]
    EPI_MUL_GELU_GRAD = 5, // C = acc * gelu'(aux[m][n])      (aux = saved pre-activation)
    EPI_MUL_RELU_MASK = 6, // C = acc * (aux[m][n] > 0)       (aux = saved ReLU output)
    EPI_BIAS_ROWADD = 7,   // C = acc + bias[n] + aux[(m % aux_rows)][n]  (patch embed: + pos_embed rows)
    EPI_ACCUM = 8,         // C += acc
    // C-ABI codes of d2s_gemm_f32_bf16io only (mapped to the two kinds above + GemmArgs::aux_bf16): the saved GELU pre-activation in bf16
    EPI_BIAS_GELU_Z16 = 9,      // as EPI_BIAS_GELU, aux_out is bf16 [M][ldc]
    EPI_MUL_GELU_GRAD_Z16 = 10, // as EPI_MUL_GELU_GRAD, aux is bf16 [M][ldaux]
};

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* aux; float* aux_out;
    long lda, ldb, ldc, ldaux;
    int M, N, K;
    int epi;
    int k_per_slice;   // reduction elements per blockIdx.z slice (multiple of BK)
    long slab_stride;  // elements between split-K slabs (0 when gridDim.z == 1)
    int aux_rows;      // EPI_BIAS_ROWADD
    int vecA, vecB;    // 16-byte vector loads allowed for A / B
    // output row remap (patch embed writes token t of image b to row b*(T+1)+1+t): out_row = m + m / rows_per_img * skip + skip0
    int remap_rows_per_img; int remap_skip;
    int stagger;       // unused (a first-round stagger of co-resident workgroups was measured slower)
    int vec_epilogue;  // 16-byte LDS-staged epilogue allowed (see epilogue_vec_ok)
    float* colsum;     // TN only: column sums of the A operand (bias gradient), one slab of M floats per K slice; may be null
    int colsum_accumulate;
    // bf16 mode only (d2s_gemm_f32_bf16io): a16 = the A operand already rounded to bf16, dense [M][K] (K % 32 == 0) - no conversion pass;
    // c16 = where to put a bf16 copy of the result, dense [M][N] (N % 32 == 0): the next GEMM's a16
    const void* a16; void* c16;
    const void* b16;   // bf16 mode: the B operand already in bf16, [N][K] k-contiguous whatever the layout (a cached weight, or W^T for dgrad)
    int aux_bf16;      // bf16 mode: aux_out (EPI_BIAS_GELU) / aux (EPI_MUL_GELU_GRAD) point at bf16, not float
};

typedef __bf16 bf16x4_epi __attribute__((ext_vector_type(4)));

template <int EPI, int MT, int NT>
__device__ __forceinline__ void store_tile_out(const GemmArgs& p, float* __restrict__ Cb, const f32x16 (&acc)[MT][NT], int mbase,
                                               int nbase, int half) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32;
        if (n >= p.N) continue;
        float bias = 0.f;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_ROWADD)
            bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + mt * 32 + mfma32_row(r, half);
                if (m >= p.M) continue;
                float v = acc[mt][nt][r];
                long orow = m;
                if (EPI == EPI_BIAS_ROWADD && p.remap_rows_per_img > 0)
                    orow = (long)m + (long)(m / p.remap_rows_per_img) * p.remap_skip + p.remap_skip;
                float* cp = Cb + orow * p.ldc + n;
                if (EPI == EPI_BIAS) v += bias;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
                if (EPI == EPI_BIAS_GELU) {
                    v += bias;
                    if (p.aux_out && p.aux_bf16) reinterpret_cast<__bf16*>(p.aux_out)[(long)m * p.ldc + n] = (__bf16)v;
                    else if (p.aux_out) p.aux_out[(long)m * p.ldc + n] = v;
                    v = gelu_erf(v);
                }
                if (EPI == EPI_BIAS_RESID) v += bias + p.aux[(long)m * p.ldaux + n];
                if (EPI == EPI_MUL_GELU_GRAD)
                    v *= gelu_erf_grad(p.aux_bf16 ? (float)reinterpret_cast<const __bf16*>(p.aux)[(long)m * p.ldaux + n] : p.aux[(long)m * p.ldaux + n]);
                if (EPI == EPI_MUL_RELU_MASK) v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
                if (EPI == EPI_BIAS_ROWADD) v += bias + p.aux[(long)(m % p.aux_rows) * p.ldaux + n];
                if (EPI == EPI_ACCUM) v += *cp;
                *cp = v;
            }
        }
    }
}

// Same epilogue, but each wave first parks 16 rows of its accumulator sub-tile in a wave-private LDS buffer [16][WN + 4] and
// reads them back row-wise, so that global traffic is 16-byte and row-contiguous (WN x 4 B per row) for the output AND for the
// residual / activation-gradient operand: 4x fewer memory instructions than the dword form.  (The per-workgroup store tail was
// measured at 8-12 us with dword stores - latency bound, not bandwidth bound - and about half that in this form.)  Requires
// N % 4 == 0, ldc % 4 == 0 and 16-byte aligned bases; `stage_wave` points at epi_stage_floats(NT) floats of LDS per wave that
// the main loop no longer uses (callers barrier after their last LDS read).
__host__ __device__ constexpr int epi_stage_floats(int NT) { return 16 * (NT * 32 + 4); }

template <int EPI, int MT, int NT, bool C16 = false>
__device__ __forceinline__ void store_tile_out_lds(const GemmArgs& p, float* __restrict__ Cb, const f32x16 (&acc)[MT][NT], int mrow0,
                                                   int ncol0, int lane, float* __restrict__ stage_wave) {
    constexpr int WN = NT * 32, PITCH = WN + 4, LPR = WN / 4, RPI = 64 / LPR, ITERS = 16 / RPI;
    const int half = lane >> 5, l31 = lane & 31;
    const int rr = lane / LPR, c4 = (lane % LPR) * 4;
    const int n = ncol0 + c4;
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_ROWADD;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS && p.bias && n < p.N) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8)
                    stage_wave[((r8 & 3) + 8 * (r8 >> 2) + 4 * half) * PITCH + nt * 32 + l31] = acc[mt][nt][h2 * 8 + r8];
            // same-wave LDS accesses complete in order: no barrier needed for the wave-private buffer
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int row = it * RPI + rr;
                const int m = mrow0 + mt * 32 + h2 * 16 + row;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage_wave + row * PITCH + c4);
                if (m >= p.M || n >= p.N) continue;
                long orow = m;
                if (EPI == EPI_BIAS_ROWADD && p.remap_rows_per_img > 0)
                    orow = (long)m + (long)(m / p.remap_rows_per_img) * p.remap_skip + p.remap_skip;
                float* cp = Cb + orow * p.ldc + n;
                if (HAS_BIAS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bias[j];
                }
                if (EPI == EPI_BIAS_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (EPI == EPI_BIAS_GELU) {
                    if (p.aux_out && p.aux_bf16) {
                        bf16x4_epi zh;
#pragma unroll
                        for (int j = 0; j < 4; ++j) zh[j] = (__bf16)v[j];
                        *reinterpret_cast<bf16x4_epi*>(reinterpret_cast<__bf16*>(p.aux_out) + (long)m * p.ldc + n) = zh;
                    } else if (p.aux_out) {
                        *reinterpret_cast<f32x4*>(p.aux_out + (long)m * p.ldc + n) = v;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                }
                if (EPI == EPI_BIAS_RESID || EPI == EPI_MUL_GELU_GRAD || EPI == EPI_MUL_RELU_MASK) {
                    f32x4 a;
                    if (EPI == EPI_MUL_GELU_GRAD && p.aux_bf16) {
                        const bf16x4_epi zh = *reinterpret_cast<const bf16x4_epi*>(reinterpret_cast<const __bf16*>(p.aux) + (long)m * p.ldaux + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[j] = (float)zh[j];
                    } else {
                        a = *reinterpret_cast<const f32x4*>(p.aux + (long)m * p.ldaux + n);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (EPI == EPI_BIAS_RESID) v[j] += a[j];
                        if (EPI == EPI_MUL_GELU_GRAD) v[j] *= gelu_erf_grad(a[j]);
                        if (EPI == EPI_MUL_RELU_MASK) v[j] = a[j] > 0.f ? v[j] : 0.f;
                    }
                }
                if (EPI == EPI_BIAS_ROWADD) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + (long)(m % p.aux_rows) * p.ldaux + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += a[j];
                }
                if (EPI == EPI_ACCUM) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += o[j];
                }
                if (!C16 || Cb) *reinterpret_cast<f32x4*>(cp) = v;
                if constexpr (C16) {
                    if (p.c16) {
                        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                        bf16x4_t h;
#pragma unroll
                        for (int j = 0; j < 4; ++j) h[j] = (__bf16)v[j];
                        *reinterpret_cast<bf16x4_t*>(static_cast<__bf16*>(p.c16) + orow * p.N + n) = h;
                    }
                }
            }
        }
    }
}

// The same LDS-staged 16-byte epilogue for accumulators of v_mfma_f32_16x16x4_f32 tiles: acc[mt][nt] is the 16x16 block (mt, nt) of the
// wave's (MT*16) x (NT*16) sub-tile; register r of lane l holds row 4 * (l >> 4) + r, column l & 15 of it.  One 16-row slab (all NT column
// blocks of one mt) is parked at a time; the read-back / global part is identical to the 32x32 form.
template <int EPI, int MT, int NT>
__device__ __forceinline__ void store_tile_out_lds16(const GemmArgs& p, float* __restrict__ Cb, const f32x4 (&acc)[MT][NT], int mrow0,
                                                     int ncol0, int lane, float* __restrict__ stage_wave) {
    constexpr int WN = NT * 16, PITCH = WN + 4, LPR = WN / 4, RPI = 64 / LPR, ITERS = 16 / RPI;
    const int lq = lane >> 4, l15 = lane & 15;
    const int rr = lane / LPR, c4 = (lane % LPR) * 4;
    const int n = ncol0 + c4;
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_ROWADD;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS && p.bias && n < p.N) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage_wave[(4 * lq + r) * PITCH + nt * 16 + l15] = acc[mt][nt][r];
        // same-wave LDS accesses complete in order: no barrier needed for the wave-private buffer
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = it * RPI + rr;
            const int m = mrow0 + mt * 16 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage_wave + row * PITCH + c4);
            if (m >= p.M || n >= p.N) continue;
            long orow = m;
            if (EPI == EPI_BIAS_ROWADD && p.remap_rows_per_img > 0)
                orow = (long)m + (long)(m / p.remap_rows_per_img) * p.remap_skip + p.remap_skip;
            float* cp = Cb + orow * p.ldc + n;
            if (HAS_BIAS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bias[j];
            }
            if (EPI == EPI_BIAS_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (EPI == EPI_BIAS_GELU) {
                if (p.aux_out) *reinterpret_cast<f32x4*>(p.aux_out + (long)m * p.ldc + n) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            }
            if (EPI == EPI_BIAS_RESID || EPI == EPI_MUL_GELU_GRAD || EPI == EPI_MUL_RELU_MASK) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + (long)m * p.ldaux + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (EPI == EPI_BIAS_RESID) v[j] += a[j];
                    if (EPI == EPI_MUL_GELU_GRAD) v[j] *= gelu_erf_grad(a[j]);
                    if (EPI == EPI_MUL_RELU_MASK) v[j] = a[j] > 0.f ? v[j] : 0.f;
                }
            }
            if (EPI == EPI_BIAS_ROWADD) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + (long)(m % p.aux_rows) * p.ldaux + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += a[j];
            }
            if (EPI == EPI_ACCUM) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += o[j];
            }
            *reinterpret_cast<f32x4*>(cp) = v;
        }
    }
}
template <int MT, int NT>
__device__ __forceinline__ void store_tile_dispatch_lds16(int epi, const GemmArgs& p, float* Cb, const f32x4 (&acc)[MT][NT], int mr, int nc,
                                                          int lane, float* stage) {
    switch (epi) {
        case EPI_BIAS: store_tile_out_lds16<EPI_BIAS, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_RELU: store_tile_out_lds16<EPI_BIAS_RELU, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_GELU: store_tile_out_lds16<EPI_BIAS_GELU, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_RESID: store_tile_out_lds16<EPI_BIAS_RESID, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_MUL_GELU_GRAD: store_tile_out_lds16<EPI_MUL_GELU_GRAD, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_MUL_RELU_MASK: store_tile_out_lds16<EPI_MUL_RELU_MASK, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_ROWADD: store_tile_out_lds16<EPI_BIAS_ROWADD, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_ACCUM: store_tile_out_lds16<EPI_ACCUM, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
        default: store_tile_out_lds16<EPI_NONE, MT, NT>(p, Cb, acc, mr, nc, lane, stage); break;
    }
}

// resolve the (wave-uniform) epilogue kind once; each kind has its own straight-line store loop
template <int MT, int NT, bool C16 = false>
__device__ __forceinline__ void store_tile_dispatch_lds(int epi, const GemmArgs& p, float* Cb, const f32x16 (&acc)[MT][NT], int mr, int nc,
                                                        int lane, float* stage) {
    switch (epi) {
        case EPI_BIAS: store_tile_out_lds<EPI_BIAS, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_RELU: store_tile_out_lds<EPI_BIAS_RELU, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_GELU: store_tile_out_lds<EPI_BIAS_GELU, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_RESID: store_tile_out_lds<EPI_BIAS_RESID, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_MUL_GELU_GRAD: store_tile_out_lds<EPI_MUL_GELU_GRAD, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_MUL_RELU_MASK: store_tile_out_lds<EPI_MUL_RELU_MASK, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_BIAS_ROWADD: store_tile_out_lds<EPI_BIAS_ROWADD, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        case EPI_ACCUM: store_tile_out_lds<EPI_ACCUM, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
        default: store_tile_out_lds<EPI_NONE, MT, NT, C16>(p, Cb, acc, mr, nc, lane, stage); break;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// can the 16-byte epilogue be used for this problem?
inline bool epilogue_vec_ok(const GemmArgs& p) {
    bool ok = (p.N % 4 == 0) && (p.ldc % 4 == 0) && aligned16(p.C);
    if (p.bias) ok = ok && aligned16(p.bias);
    if (p.aux) ok = ok && (p.ldaux % 4 == 0) && aligned16(p.aux);
    if (p.aux_out) ok = ok && aligned16(p.aux_out);
    return ok;
}

}  // namespace d2s_gemm
