// Bare fp32 MFMA loops on random operands: which shape sustains more FLOP/s (and which clock) on this chip?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_f32.hip -o gpurun_out/mfma_shape_f32 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // 16 MFMAs of 32x32x2 per trip = 65536 FLOP per wave-trip... x4 accumulators
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 3], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 3], b[u], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 2) & 3], b[(u + 3) & 3], acc[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {          // 32 MFMAs of 16x16x4 per trip = the same 65536... FLOP as above
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + u) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    // Residency sweep: dynamic LDS per workgroup caps the workgroups a CU can hold (160 KB per CU): 96 KB -> 1 (one wave per SIMD),
    // 64 KB -> 2, 48 KB -> 3, 36 KB -> 4.  The grid always supplies 4 workgroups per CU of work.
    const int blocks = 256 * 4, iters = 20000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<float> h(1 << 20); srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2e-3f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned long long> hc(blocks * 2);
    hipFuncSetAttribute((const void*)loop_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)loop_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds_kb : {96, 64, 48, 36}) {
        for (int shape : {32, 16, 32, 16}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), lds_kb * 1024, 0, in, out, iters, clk);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), lds_kb * 1024, 0, in, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += 100.0 * hc[2 * i] / hc[2 * i + 1]; mhz /= blocks;
            const double flop = (double)blocks * 4 /*waves*/ * iters * 16 * 4096.0;   // per trip: 16 x 32x32x2 (4096 FLOP) == 32 x 16x16x4 (2048 FLOP)
            const double peak_at_clock = 256.0 * 4 * 64 * mhz * 1e6 / 1e12;
            printf("workgroups/CU %d (LDS %3d KB)  shape %2dx%-2d: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %5.0f MHz  -> %.3f of the issue peak at that clock\n",
                   lds_kb == 96 ? 1 : lds_kb == 64 ? 2 : lds_kb == 48 ? 3 : 4, lds_kb, shape, shape, ms, flop / ms / 1e9, mhz, flop / ms / 1e9 / peak_at_clock);
        }
    }
    return 0;
}
