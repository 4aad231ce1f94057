// Which part of the fp32 GEMM's data movement costs the clock?  The main loop of gemm_f32_rk_kernel<0,128,128> (128x128 tile, 4 waves of
// 64x64, BK 16, [row][k] LDS image, double-buffered, 3 workgroups per CU) with parts switched off at compile time:
//   LOADS: global_load_dwordx4 -> VGPR -> ds_write_b128 of the next K-step (off: the LDS images are filled once and re-used)
//   READS: ds_read_b128 fragment reads every K-step (off: the fragments are read once and re-used)
// All variants issue the same MFMAs on N(0,1) data.  A bare MFMA loop holds 2.38 GHz / 155 TFLOP/s (profiles/r03_b_mfma_power_real_data.txt).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm_power_f32.hip -o tools/micro/gemm_power_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 16, BM = 128, BN = 128;

template <bool LOADS, bool READS>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                            unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* As = smem;
    float* Bs = smem + 2 * BK * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int nbn = N / BN, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int nk = K / BK;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    long offa[2], offb[2];
    for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256;
        offa[i] = (long)(bm * BM + (f >> 2)) * K + (f & 3) * 4;
        offb[i] = (long)(bn * BN + (f >> 2)) * K + (f & 3) * 4;
    }
    f32x4 ra[2], rb[2];
    auto ld = [&](int k0) { for (int i = 0; i < 2; ++i) { ra[i] = *reinterpret_cast<const f32x4*>(A + k0 + offa[i]); rb[i] = *reinterpret_cast<const f32x4*>(B + k0 + offb[i]); } };
    auto st = [&](int buf) {
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * 256, row = f >> 2, kq = f & 3;
            *reinterpret_cast<f32x4*>(&As[buf * BK * BM + row * BK + ((kq ^ ((row >> 2) & 3)) << 2)]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[buf * BK * BN + row * BK + ((kq ^ ((row >> 2) & 3)) << 2)]) = rb[i];
        }
    };
    ld(0); st(0); ld(BK); st(1);
    __syncthreads();
    int a_off[2][2], b_off[2][2];
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 2; ++c) {
        const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
        a_off[i][c] = ra_ * BK + (((2 * half + c) ^ ((ra_ >> 2) & 3)) << 2);
        b_off[i][c] = rb_ * BK + (((2 * half + c) ^ ((rb_ >> 2) & 3)) << 2);
    }
    f32x4 av[2][2], bv[2][2];
    for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) { av[i][c] = *reinterpret_cast<const f32x4*>(As + a_off[i][c]); bv[i][c] = *reinterpret_cast<const f32x4*>(Bs + b_off[i][c]); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (LOADS) ld(min(kt + 1, nk - 1) * BK);
        if (READS) {
            const float* Ac = As + cur * BK * BM;
            const float* Bc = Bs + cur * BK * BN;
            for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) { av[i][c] = *reinterpret_cast<const f32x4*>(Ac + a_off[i][c]); bv[i][c] = *reinterpret_cast<const f32x4*>(Bc + b_off[i][c]); }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s >> 2][s & 3], bv[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (LOADS && kt + 1 < nk) st(cur ^ 1);
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    C[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}


// ---- variant: LDS-DMA staging (global_load_lds_dwordx4, no VGPR / ds_write), 3-slab ring, counted vmcnt + raw barrier, 3 workgroups per CU ----
__global__ __launch_bounds__(256, 3) void k_dma(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                unsigned long long* clk) {
    constexpr int SLAB = BK * (BM + BN);      // floats per ring slot (16 KB)
    __shared__ __attribute__((aligned(1024))) float smem[3 * SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int nbn = N / BN, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int nk = K / BK;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // wave w issues 4 DMA instructions per K-step: j = 4 w + i; j < 8: A rows 16 j .. 16 j + 15, else B rows 16 (j - 8) ..; lane l -> row + l / 4,
    // LDS chunk l % 4 holds source chunk (l % 4) ^ ((row >> 2) & 3)  (the swizzle goes on the SOURCE address, the LDS image stays lane-linear)
    const float* src[4];
    int dst[4];
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i, isb = j >= 8, row = (isb ? j - 8 : j) * 16 + (lane >> 2), ch = (lane & 3) ^ ((row >> 2) & 3);
        src[i] = (isb ? B + (long)(bn * BN + row) * K : A + (long)(bm * BM + row) * K) + ch * 4;
        dst[i] = (isb ? BK * BM : 0) + (isb ? j - 8 : j) * 16 * BK;      // wave-uniform float offset inside a slab
    }
    auto issue = [&](int kt, int buf) {
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(smem + buf * SLAB + dst[i]), 16, 0, 0);
    };
    int a_off[2][2], b_off[2][2];
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 2; ++c) {
        const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
        a_off[i][c] = ra_ * BK + (((2 * half + c) ^ ((ra_ >> 2) & 3)) << 2);
        b_off[i][c] = BK * BM + rb_ * BK + (((2 * half + c) ^ ((rb_ >> 2) & 3)) << 2);
    }
    issue(0, 0);
    issue(1, 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // slab kt landed (kt + 1 may be in flight)
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);      // into the slot every wave finished reading in the previous step
        const float* S = smem + buf * SLAB;
        f32x4 av[2][2], bv[2][2];
        for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) { av[i][c] = *reinterpret_cast<const f32x4*>(S + a_off[i][c]); bv[i][c] = *reinterpret_cast<const f32x4*>(S + b_off[i][c]); }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s >> 2][s & 3], bv[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        buf = buf == 2 ? 0 : buf + 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    C[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- variant: 32-deep K-steps (every global load instruction moves full 128-byte lines: 8 rows x 128 B instead of 16 rows x 64 B), register
// staged, 64 KB of LDS -> 2 workgroups per CU ----
__global__ __launch_bounds__(256, 2) void k_bk32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                 unsigned long long* clk) {
    constexpr int BK2 = 32;
    extern __shared__ __attribute__((aligned(16))) float dsm[];      // [2][BM + BN][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
    const int nbn = N / BN, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int nk = K / BK2;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // thread t, i = 0..3: flat chunk f = t + 256 i of the A (and B) tile: row = f / 8, 16-byte chunk kq = f % 8
    long offa[4], offb[4];
    int soff[4];
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, row = f >> 3, kq = f & 7;
        offa[i] = (long)(bm * BM + row) * K + kq * 4;
        offb[i] = (long)(bn * BN + row) * K + kq * 4;
        soff[i] = row * BK2 + ((kq ^ (row & 7)) << 2);
    }
    f32x4 ra[4], rb[4];
    auto ld = [&](int k0) { for (int i = 0; i < 4; ++i) { ra[i] = *reinterpret_cast<const f32x4*>(A + k0 + offa[i]); rb[i] = *reinterpret_cast<const f32x4*>(B + k0 + offb[i]); } };
    auto st = [&](int buf) {
        float* S = dsm + buf * BK2 * (BM + BN);
        for (int i = 0; i < 4; ++i) { *reinterpret_cast<f32x4*>(S + soff[i]) = ra[i]; *reinterpret_cast<f32x4*>(S + BK2 * BM + soff[i]) = rb[i]; }
    };
    ld(0); st(0);
    __syncthreads();
    // MFMA step q of the 32-deep step multiplies k = q (lanes 0-31) and 16 + q (lanes 32-63): chunk = 4 half + c, c = 0..3
    int a_off[2][4], b_off[2][4];
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 4; ++c) {
        const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
        a_off[i][c] = ra_ * BK2 + (((4 * half + c) ^ (ra_ & 7)) << 2);
        b_off[i][c] = BK2 * BM + rb_ * BK2 + (((4 * half + c) ^ (rb_ & 7)) << 2);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        ld(min(kt + 1, nk - 1) * BK2);
        const float* S = dsm + cur * BK2 * (BM + BN);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 av[2], bv[2];
            for (int i = 0; i < 2; ++i) { av[i] = *reinterpret_cast<const f32x4*>(S + a_off[i][c]); bv[i] = *reinterpret_cast<const f32x4*>(S + b_off[i][c]); }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) st(cur ^ 1);
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    C[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename KF>
void run_k(const char* name, KF kern, size_t dyn_lds, const float* A, const float* B, float* C, int M, int N, int K, unsigned long long* clk) {
    const int blocks = (M / BM) * (N / BN);
    std::vector<unsigned long long> hc(blocks * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), dyn_lds, 0, A, B, C, M, N, K, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
    double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += 100.0 * hc[2 * i] / hc[2 * i + 1]; mhz /= blocks;
    printf("M %6d N %5d K %5d  %-34s %8.1f us  %7.1f TFLOP/s  in-loop clock %5.0f MHz\n", M, N, K, name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, mhz);
}

template <bool L, bool R>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, unsigned long long* clk) {
    const int blocks = (M / BM) * (N / BN);
    std::vector<unsigned long long> hc(blocks * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<L, R>), dim3(blocks), dim3(256), 0, 0, A, B, C, M, N, K, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
    double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += 100.0 * hc[2 * i] / hc[2 * i + 1]; mhz /= blocks;
    printf("M %6d N %5d K %5d  %-34s %8.1f us  %7.1f TFLOP/s  in-loop clock %5.0f MHz\n", M, N, K, name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, mhz);
}

int main() {
    const int shapes[3][3] = {{24576, 1536, 4096}, {24576, 1536, 384}, {24576, 384, 1536}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        float *A, *B, *C; unsigned long long* clk;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)(M / BM) * (N / BN) * 256 * 4); hipMalloc(&clk, (size_t)(M / BM) * (N / BN) * 16);
        std::vector<float> h((size_t)M * K); srand(1);
        for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 3.46f;
        hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
        run<true, true>("full (loads + LDS reads + MFMA)", A, B, C, M, N, K, clk);
        run<false, true>("no global loads / ds_write", A, B, C, M, N, K, clk);
        run<true, false>("no LDS fragment reads", A, B, C, M, N, K, clk);
        run<false, false>("MFMA + barrier only", A, B, C, M, N, K, clk);
        run_k("LDS-DMA ring (3 slabs, 3 WG/CU)", k_dma, 0, A, B, C, M, N, K, clk);
        hipFuncSetAttribute((const void*)k_bk32, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        run_k("BK 32 full-line loads (2 WG/CU)", k_bk32, 64 * 1024, A, B, C, M, N, K, clk);
        hipFree(A); hipFree(B); hipFree(C); hipFree(clk);
    }
    return 0;
}
