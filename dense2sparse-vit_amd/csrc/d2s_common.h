// Shared device helpers for the d2s HIP library (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define D2S_OK 0
#define D2S_ERR_ARG (-1)       // bad shape / null pointer / unsupported size
#define D2S_ERR_WORKSPACE (-2) // workspace too small
#define D2S_ERR_LAUNCH (-3)    // hipGetLastError() after launch

#define D2S_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int d2s_check_launch() { return hipGetLastError() == hipSuccess ? D2S_OK : D2S_ERR_LAUNCH; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 32 lanes of each half-wave (lanes 0-31 / 32-63 separately)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 32x32x2 f32 MFMA.  Lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// C/D register r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// XCD-aware remap of a 2-D grid.  Workgroups are handed to the 8 XCDs round-robin in linear id order (x fastest), so the blockIdx.x tiles
// of one blockIdx.y row - the query (key) blocks of one attention head, which all stream the same K / V (Q / dO) panels - would land on
// different XCDs, each pulling its own copy of the panels into its L2.  This maps the hardware id to a virtual (bx, by) such that every
// XCD owns a contiguous range of the x-fastest linear order (bijective, same construction as the GEMM tile order).
__device__ __forceinline__ void xcd_remap_2d(int& bx, int& by) {
#ifdef D2S_NO_XCD_REMAP_2D      // diagnostic build for the A/B (tools/attn_bench.py with D2S_LIB_PATH)
    bx = (int)blockIdx.x; by = (int)blockIdx.y;
    return;
#endif
    const int gx = (int)gridDim.x, total = gx * (int)gridDim.y;
    const int L = (int)blockIdx.y * gx + (int)blockIdx.x;
    const int xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
    const int V = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = V / gx;
    bx = V - by * gx;
}
