// Is the fp32 matrix pipe itself power-limited on real data?  Bare v_mfma_f32_32x32x2_f32 loops, operands in registers, no LDS / global
// traffic inside the loop, one workgroup of 4 waves per SIMD set (residency 1, 2, 3 workgroups per CU):
//   "same"    8 operand registers of tiny values re-used every trip (tools/micro/mfma_shape_f32.hip: 2.38 GHz, 155 TFLOP/s)
//   "varied"  32 operand registers of N(0,1)-scale values, every MFMA multiplies another pair: the operand toggling of a real GEMM
//   "zero"    all-zero operands
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power_f32.hip -o tools/micro/mfma_power_f32 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOP>      // NOP distinct A and B operand registers
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[NOP], b[NOP];
#pragma unroll
    for (int i = 0; i < NOP; ++i) { a[i] = in[(tid * 2 * NOP + i) & 0xFFFFF]; b[i] = in[(tid * 2 * NOP + NOP + i) & 0xFFFFF]; }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {          // 64 MFMAs per trip, 4 independent accumulators (the 2x2 tile of a GEMM wave)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u) % NOP], b[(2 * u) % NOP], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u) % NOP], b[(2 * u + 1) % NOP], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u + 1) % NOP], b[(2 * u) % NOP], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * u + 1) % NOP], b[(2 * u + 1) % NOP], acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int iters = 6000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&clk, 256 * 4 * 16);
    std::vector<float> h(1 << 20);
    hipFuncSetAttribute((const void*)loop_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)loop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int data = 0; data < 3; ++data) {       // 0 = N(0,1)-scale uniform, 1 = tiny, 2 = zero
        srand(1);
        for (auto& v : h) v = data == 2 ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * (data == 0 ? 3.46f : 2e-3f);
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int wgs : {1, 2, 3}) {
            const int lds_kb = wgs == 1 ? 96 : wgs == 2 ? 64 : 48, blocks = 256 * wgs;
            for (int nop : {2, 32}) {
                std::vector<unsigned long long> hc(blocks * 2);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (nop == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), lds_kb * 1024, 0, in, out, iters, clk);
                    else hipLaunchKernelGGL(loop_kernel<2>, dim3(blocks), dim3(256), lds_kb * 1024, 0, in, out, iters, clk);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
                double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += 100.0 * hc[2 * i] / hc[2 * i + 1]; mhz /= blocks;
                const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
                printf("data %-5s  workgroups/CU %d  %2d operand registers: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %5.0f MHz\n",
                       data == 0 ? "N(0,1)" : data == 1 ? "tiny" : "zero", wgs, 2 * nop, ms, flop / ms / 1e9, mhz);
            }
        }
    }
    return 0;
}
