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
    int stagger;       // de-synchronise the first residency round (speed only)
};

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
                    if (p.aux_out) p.aux_out[(long)m * p.ldc + n] = v;
                    v = gelu_erf(v);
                }
                if (EPI == EPI_BIAS_RESID) v += bias + p.aux[(long)m * p.ldaux + n];
                if (EPI == EPI_MUL_GELU_GRAD) v *= gelu_erf_grad(p.aux[(long)m * p.ldaux + n]);
                if (EPI == EPI_MUL_RELU_MASK) v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
                if (EPI == EPI_BIAS_ROWADD) v += bias + p.aux[(long)(m % p.aux_rows) * p.ldaux + n];
                if (EPI == EPI_ACCUM) v += *cp;
                *cp = v;
            }
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace d2s_gemm
