// BatchNorm1d over the token rows of a [R, C] activation (R = B * tokens, C channels contiguous) - the `--predictor-bn` variant of
// the mask predictor (vit_models/dynamic_vit.py:350-367 BatchNormLayer: transpose -> nn.BatchNorm1d -> transpose, i.e. per-channel
// statistics over all B * N rows).  Training mode: batch statistics (biased variance for the normalisation, unbiased for the running
// estimate, momentum 0.1 as nn.BatchNorm1d defaults); eval mode: running statistics.  HBM-bound: statistics are one pass (column
// sums of x and x^2 as float partials per row slice, combined in double), the normalisation a second pass.
#include "d2s_common.h"

namespace {

constexpr int BN_ROWS_PER_SLICE = 256;

inline int bn_slices(long R) {
    long s = (R + BN_ROWS_PER_SLICE - 1) / BN_ROWS_PER_SLICE;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

// part[slice][2][C]: column sums of a and a*b over the slice's rows (b == nullptr: a*a).  xhat_of: b is dy and a must first be
// normalised with (mean, rstd) -> sums of dy and dy * xhat (the backward statistics).
__global__ __launch_bounds__(256) void bn_colsums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd, long R, int C,
                                                         long rows_per_slice, float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_slice, r1 = min(R, r0 + rows_per_slice);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mu = mean ? mean[c] : 0.f, rs = rstd ? rstd[c] : 1.f;
        for (long r = r0 + wave; r < r1; r += 4) {
            const float av = a[r * C + c];
            if (b) {                      // backward: b = dy, a = x
                const float dy = b[r * C + c];
                s0 += dy;
                s1 += dy * (av - mu) * rs;
            } else {                      // forward: sums of x and x^2
                s0 += av;
                s1 += av * av;
            }
        }
    }
    red[0][wave][lane] = s0;
    red[1][wave][lane] = s1;
    __syncthreads();
    if (wave == 0 && c < C) {
        part[((long)blockIdx.y * 2 + 0) * C + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        part[((long)blockIdx.y * 2 + 1) * C + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}

// training statistics from the partials; updates the running estimates exactly like nn.BatchNorm1d (momentum, unbiased variance)
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ part, int slices, long R, int C, float eps,
                                                              float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int z = 0; z < slices; ++z) {
        s += (double)part[((long)z * 2 + 0) * C + c];
        q += (double)part[((long)z * 2 + 1) * C + c];
    }
    const double mu = s / (double)R;
    double var = q / (double)R - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && running_var) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                            int C, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, long total, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    y[i] = (x[i] - mean[c]) * rstd[c] * w[c] + b[c];
}

// dw / db from the backward partials; col[0][c] = sum dy / R, col[1][c] = sum dy * xhat / R for the dx pass
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ part, int slices, long R, int C,
                                                              float* __restrict__ col, float* __restrict__ dw, float* __restrict__ db,
                                                              int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int z = 0; z < slices; ++z) {
        s += (double)part[((long)z * 2 + 0) * C + c];
        q += (double)part[((long)z * 2 + 1) * C + c];
    }
    col[c] = (float)(s / (double)R);
    col[C + c] = (float)(q / (double)R);
    if (db) db[c] = accumulate ? db[c] + (float)s : (float)s;
    if (dw) dw[c] = accumulate ? dw[c] + (float)q : (float)q;
}

// dx = rstd * w * (dy - mean(dy) - xhat * mean(dy * xhat)); optionally masked by x > 0 (a ReLU whose OUTPUT is this layer's input)
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ w, const float* __restrict__ col,
                                                           float* __restrict__ dx, long total, int C, int relu_mask, int training) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float xv = x[i];
    float g;
    if (training) {
        const float xh = (xv - mean[c]) * rstd[c];
        g = rstd[c] * w[c] * (dy[i] - col[c] - xh * col[C + c]);
    } else {
        g = rstd[c] * w[c] * dy[i];       // eval mode: the statistics are constants
    }
    dx[i] = (relu_mask && !(xv > 0.f)) ? 0.f : g;
}

}  // namespace

extern "C" {

// scratch: slices * 2 * C partials + 2 * C column means of the backward
size_t d2s_batchnorm_workspace_bytes(long R, int C) { return ((size_t)bn_slices(R) * 2 * C + 2 * (size_t)C) * sizeof(float); }

// y = (x - mean) * rstd * w + b over [R, C]; training != 0: batch statistics (saved in mean / rstd, running estimates updated when
// given), else the running estimates (copied into mean / rstd for the backward).
int d2s_batchnorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, float* running_mean,
                      float* running_var, long R, int C, float eps, float momentum, int training, void* workspace,
                      size_t workspace_bytes, hipStream_t stream) {
    if (!x || !w || !b || !y || !mean || !rstd || R <= 0 || C <= 0) return D2S_ERR_ARG;
    if (!training && (!running_mean || !running_var)) return D2S_ERR_ARG;
    if (training) {
        if (!workspace || workspace_bytes < d2s_batchnorm_workspace_bytes(R, C)) return D2S_ERR_WORKSPACE;
        const int slices = bn_slices(R);
        const long rps = (R + slices - 1) / slices;
        float* part = static_cast<float*>(workspace);
        hipLaunchKernelGGL(bn_colsums_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, stream, x, static_cast<const float*>(nullptr),
                           static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), R, C, rps, part);
        hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, slices, R, C, eps, momentum, mean,
                           rstd, running_mean, running_var);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_mean, running_var, C, eps, mean, rstd);
    }
    const long total = R * C;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, mean, rstd, w, b, y, total, C);
    return d2s_check_launch();
}

// dx (always), dw / db (+)= when given.  relu_mask: dx is zeroed where x <= 0.  training must match the forward.
int d2s_batchnorm_bwd(const float* x, const float* dy, const float* w, const float* mean, const float* rstd, float* dx, float* dw,
                      float* db, int relu_mask, int accumulate, int training, long R, int C, void* workspace, size_t workspace_bytes,
                      hipStream_t stream) {
    if (!x || !dy || !w || !mean || !rstd || !dx || R <= 0 || C <= 0) return D2S_ERR_ARG;
    if (!workspace || workspace_bytes < d2s_batchnorm_workspace_bytes(R, C)) return D2S_ERR_WORKSPACE;
    const int slices = bn_slices(R);
    const long rps = (R + slices - 1) / slices;
    float* part = static_cast<float*>(workspace);
    float* col = part + (size_t)slices * 2 * C;
    hipLaunchKernelGGL(bn_colsums_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, stream, x, dy, mean, rstd, R, C, rps, part);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, slices, R, C, col, dw, db, accumulate);
    const long total = R * C;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, dy, mean, rstd, w, col, dx,
                       total, C, relu_mask, training);
    return d2s_check_launch();
}

}  // extern "C"
