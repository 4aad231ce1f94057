// Loss-side kernels of the training step (losses.py) and the fused AdamW update (mask_predictor.py optimiser).
// All small and launch/HBM-bound: one wave per row, log-sum-exp in registers, deterministic two-stage sums.
//
//   teacher_target : losses.py:76-79   mean over layers, max over heads, drop CLS column, renormalise
//   gather_renorm  : losses.py:89-90   re-gather the target through the previous stage's kept ids, renormalise
//   kl_rows        : losses.py:94-95 (KL with probability target), :196 (cross entropy), :198-203 and :220-225
//                    (KL between two log-softmaxes); writes the per-row loss and d loss / d student row
//   mask_agreement : losses.py:84-96,121-164  fraction of tokens on which two top-k masks agree
#include "d2s_common.h"

namespace {

struct RowMap {
    long rows_per_group, group_stride, row_stride, offset;
};
__device__ __forceinline__ long map_row(const RowMap& m, long r) {
    const long g = r / m.rows_per_group, t = r - g * m.rows_per_group;
    return g * m.group_stride + m.offset + t * m.row_stride;
}

// cls_attn [B,L,H,n] -> target [B,n-1]
__global__ __launch_bounds__(256) void teacher_target_kernel(const float* __restrict__ a, float* __restrict__ target, int L,
                                                             int H, int n) {
    extern __shared__ float w[];  // [n]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* ab = a + (long)blockIdx.x * L * H * n;
    float part = 0.f;
    for (int t = tid; t < n; t += 256) {
        float mx = -INFINITY;
        for (int h = 0; h < H; ++h) {
            float s = 0.f;
            for (int l = 0; l < L; ++l) s += ab[((long)l * H + h) * n + t];
            mx = fmaxf(mx, s / (float)L);
        }
        w[t] = mx;
        if (t >= 1) part += mx;
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int t = 1 + tid; t < n; t += 256) target[(long)blockIdx.x * (n - 1) + t - 1] = w[t] / tot;
}

// out[b,j] = in[b,ids[b,j]] / sum_j in[b,ids[b,j]]
__global__ __launch_bounds__(256) void gather_renorm_kernel(const float* __restrict__ in, const long long* __restrict__ ids,
                                                            float* __restrict__ out, int T, int k, int normalize) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* ib = in + (long)blockIdx.x * T;
    const long long* idb = ids + (long)blockIdx.x * k;
    float part = 0.f;
    for (int j = tid; j < k; j += 256) part += ib[idb[j]];
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float tot = normalize ? (red[0] + red[1]) + (red[2] + red[3]) : 1.0f;
    for (int j = tid; j < k; j += 256) out[(long)blockIdx.x * k + j] = normalize ? ib[idb[j]] / tot : ib[idb[j]];
}

enum KlMode : int { KL_LOGIT_TARGET = 0, KL_PROB_TARGET = 1, CE_LABEL = 2, MSE_TARGET = 3, SOFT_CE = 4 };

// one wave per row.  loss_row[r], grad[r][:] = d loss_row / d s[r][:]
template <int NE>
__global__ __launch_bounds__(256) void kl_rows_kernel(const float* __restrict__ s, RowMap sm, const float* __restrict__ t, RowMap tm,
                                                      const long long* __restrict__ t_ids, const long long* __restrict__ labels,
                                                      float* __restrict__ loss_row, float* __restrict__ grad, long rows, int C,
                                                      int mode, const float* __restrict__ row_weight) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + map_row(sm, row);
    float sv[NE], tv[NE];
    float ms = -INFINITY, mt = -INFINITY;
    const float* tr = nullptr;
    if (mode != CE_LABEL) {
        long off;
        if (t_ids)  // gathered target: group (image) from the row map, row inside the image from the id list
            off = (row / tm.rows_per_group) * tm.group_stride + tm.offset + t_ids[row] * tm.row_stride;
        else
            off = map_row(tm, row);
        tr = t + off;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + i * 64;
        sv[i] = c < C ? sr[c] : -INFINITY;
        tv[i] = (tr && c < C) ? tr[c] : ((mode == KL_PROB_TARGET || mode == MSE_TARGET || mode == SOFT_CE) ? 0.f : -INFINITY);
        ms = fmaxf(ms, sv[i]);
        if (mode == KL_LOGIT_TARGET) mt = fmaxf(mt, tv[i]);
    }
    ms = wave_max(ms);
    float es = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) es += (lane + i * 64 < C) ? expf(sv[i] - ms) : 0.f;
    es = wave_sum(es);
    const float lse_s = ms + logf(es);
    float loss = 0.f;
    float* gr = grad ? grad + row * C : nullptr;
    if (mode == KL_LOGIT_TARGET) {
        mt = wave_max(mt);
        float et = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) et += (lane + i * 64 < C) ? expf(tv[i] - mt) : 0.f;
        et = wave_sum(et);
        const float lse_t = mt + logf(et);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float lt = tv[i] - lse_t, ls = sv[i] - lse_s, pt = expf(lt);
                loss += pt * (lt - ls);
                if (gr) gr[c] = expf(ls) - pt;
            }
        }
    } else if (mode == KL_PROB_TARGET) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) tsum += tv[i];
        tsum = wave_sum(tsum);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float ls = sv[i] - lse_s;
                loss += tv[i] * (logf(tv[i]) - ls);
                if (gr) gr[c] = expf(ls) * tsum - tv[i];
            }
        }
    } else if (mode == SOFT_CE) {         // soft-target cross entropy (timm SoftTargetCrossEntropy under mixup, losses.py:170-172):
        float tsum = 0.f;                 // -sum_c t_c log_softmax(s)_c; zero targets contribute nothing
#pragma unroll
        for (int i = 0; i < NE; ++i) tsum += tv[i];
        tsum = wave_sum(tsum);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float ls = sv[i] - lse_s;
                loss -= tv[i] * ls;
                if (gr) gr[c] = expf(ls) * tsum - tv[i];
            }
        }
    } else if (mode == MSE_TARGET) {      // sum of squared differences between the raw scores and the target row (losses.py:61-73)
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float d = sv[i] - tv[i];
                loss += d * d;
                if (gr) gr[c] = 2.f * d;
            }
        }
    } else {
        const long long y = labels[row];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float ls = sv[i] - lse_s;
                if (c == y) loss -= ls;
                if (gr) gr[c] = expf(ls) - (c == y ? 1.f : 0.f);
            }
        }
    }
    loss = wave_sum(loss);
    if (row_weight) {       // weighted rows (dynamic keep ratio: mask / count restricts the mean to the kept tokens)
        const float w = row_weight[row];
        loss *= w;
        if (gr) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int c = lane + i * 64;
                if (c < C) gr[c] *= w;      // same lane wrote gr[c] above
            }
        }
    }
    if (lane == 0) loss_row[row] = loss;
}

// out[0] = scale * sum(v[0..n)) with a fixed summation tree (single workgroup)
__global__ __launch_bounds__(256) void sum_scalar_kernel(const float* __restrict__ v, long n, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (long i = tid; i < n; i += 256) s += v[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) out[0] = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

// y[i] = a * x[i]   (scale a saved unit gradient by the upstream scalar gradient, read from device memory)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ gscalar,
                                                              float scale, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * (scale * gscalar[0]);
}

// out = g * act'(z): kind 0 = exact-erf GELU with z the pre-activation, kind 1 = ReLU with z the ReLU output
__global__ __launch_bounds__(256) void act_grad_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                       float* __restrict__ out, long n, int kind) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = kind == 0 ? g[i] * gelu_erf_grad(z[i]) : (z[i] > 0.f ? g[i] : 0.f);
}

// agree[b] = number of token positions on which the two top-k masks (given as id lists) coincide
__global__ __launch_bounds__(256) void mask_agreement_kernel(const long long* __restrict__ a, const long long* __restrict__ b,
                                                             int T, int k, float* __restrict__ agree) {
    extern __shared__ int m[];  // [T]
    __shared__ int red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < T; t += 256) m[t] = 0;
    __syncthreads();
    for (int j = tid; j < k; j += 256) m[a[(long)blockIdx.x * k + j]] = 1;
    __syncthreads();
    int both = 0;
    for (int j = tid; j < k; j += 256) both += m[b[(long)blockIdx.x * k + j]];
    both = wave_sum_i(both);
    if (lane == 0) red[wave] = both;
    __syncthreads();
    if (tid == 0) {
        const int inter = red[0] + red[1] + red[2] + red[3];
        agree[blockIdx.x] = (float)(T - 2 * (k - inter));
    }
}

// Fused AdamW over a flat parameter arena.  Chunk c covers elements [c*CH, (c+1)*CH); desc[c] = {lr, weight_decay,
// active} selects the hyper-parameters of the tensor the chunk belongs to (tensors are padded to CH in the arena).
// torch.optim.AdamW semantics: p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
constexpr int CH = 1024;
struct ChunkDesc { float lr, wd; int active; int pad; };
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const ChunkDesc* __restrict__ desc, float b1, float b2,
                                                    float eps, float bc1, float bc2_sqrt, float grad_scale,
                                                    int* __restrict__ chunk_steps) {
    const ChunkDesc d = desc[blockIdx.x];
    if (!d.active) return;
    // torch.optim.AdamW keeps state['step'] PER PARAMETER and only advances it when the parameter has a gradient: a tensor that was
    // frozen during the warm-up epochs starts its bias correction at t = 1 when it is first updated.  chunk_steps[c] is that counter
    // for the tensor chunk c belongs to (advanced here, for active chunks only); without it the global step is used for every chunk.
    if (chunk_steps) {
        __shared__ float bc[2];
        const int t = chunk_steps[blockIdx.x] + 1;
        if (threadIdx.x == 0) {
            bc[0] = (float)(1.0 - pow((double)b1, (double)t));
            bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
        }
        __syncthreads();    // every thread has read the old counter before thread 0 replaces it
        bc1 = bc[0];
        bc2_sqrt = bc[1];
        if (threadIdx.x == 0) chunk_steps[blockIdx.x] = t;
    }
    const long base = (long)blockIdx.x * CH + threadIdx.x * 4;
    f32x4 pv = *reinterpret_cast<f32x4*>(p + base);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + base);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + base);
    const float step = d.lr / bc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float gj = gv[j] * grad_scale;
        pv[j] *= 1.f - d.lr * d.wd;
        mv[j] = b1 * mv[j] + (1.f - b1) * gj;
        vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
        pv[j] -= step * mv[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
    }
    *reinterpret_cast<f32x4*>(p + base) = pv;
    *reinterpret_cast<f32x4*>(m + base) = mv;
    *reinterpret_cast<f32x4*>(v + base) = vv;
}

}  // namespace

extern "C" {

int d2s_teacher_target(const float* cls_attn, float* target, int B, int L, int H, int n, hipStream_t stream) {
    if (!cls_attn || !target || B <= 0 || L <= 0 || H <= 0 || n <= 1) return D2S_ERR_ARG;
    hipLaunchKernelGGL(teacher_target_kernel, dim3(B), dim3(256), (size_t)n * sizeof(float), stream, cls_attn, target, L, H, n);
    return d2s_check_launch();
}

int d2s_gather_renorm(const float* in, const long long* ids, float* out, int B, int T, int k, int normalize, hipStream_t stream) {
    if (!in || !ids || !out || B <= 0 || T <= 0 || k <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(gather_renorm_kernel, dim3(B), dim3(256), 0, stream, in, ids, out, T, k, normalize);
    return d2s_check_launch();
}

// mode 0: KL(log_softmax(s) || log_softmax(t)) per row; 1: KL with t given as probabilities; 2: cross entropy with labels;
// 3: sum of squared differences s - t (the mse mask loss); 4: soft-target cross entropy -sum t log_softmax(s) (mixup labels).
// s rows through (s_*) row map, t rows through (t_*) row map plus optional t_ids[r] * t_row_stride.
// loss_row [rows]; grad [rows, C] contiguous (may be null).  C <= 1024.
int d2s_kl_rows(const float* s, long s_rpg, long s_gs, long s_rs, long s_off, const float* t, long t_rpg, long t_gs, long t_rs,
                long t_off, const long long* t_ids, const long long* labels, float* loss_row, float* grad, long rows, int C,
                int mode, const float* row_weight, hipStream_t stream) {
    if (!s || !loss_row || rows <= 0 || C <= 0 || C > 1024 || mode < 0 || mode > 4) return D2S_ERR_ARG;
    if (mode == 2 ? !labels : !t) return D2S_ERR_ARG;
    RowMap sm{s_rpg, s_gs, s_rs, s_off}, tm{t_rpg > 0 ? t_rpg : 1, t_gs, t_rs, t_off};
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int ne = (C + 63) / 64;
#define D2S_KL(NE) hipLaunchKernelGGL(kl_rows_kernel<NE>, grid, block, 0, stream, s, sm, t, tm, t_ids, labels, loss_row, grad, rows, C, mode, row_weight)
    if (ne <= 1) D2S_KL(1);
    else if (ne <= 2) D2S_KL(2);
    else if (ne <= 4) D2S_KL(4);
    else if (ne <= 8) D2S_KL(8);
    else D2S_KL(16);
#undef D2S_KL
    return d2s_check_launch();
}

int d2s_sum_scalar(const float* v, long n, float scale, float* out, hipStream_t stream) {
    if (!v || !out || n <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, stream, v, n, scale, out);
    return d2s_check_launch();
}

int d2s_scale_by_scalar(const float* x, const float* gscalar, float scale, float* y, long n, hipStream_t stream) {
    if (!x || !gscalar || !y || n <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, gscalar, scale, y, n);
    return d2s_check_launch();
}

int d2s_mask_agreement(const long long* ids_a, const long long* ids_b, int B, int T, int k, float* agree, hipStream_t stream) {
    if (!agree || B <= 0 || T <= 0 || k < 0 || k > T || (k > 0 && (!ids_a || !ids_b))) return D2S_ERR_ARG;
    hipLaunchKernelGGL(mask_agreement_kernel, dim3(B), dim3(256), (size_t)T * sizeof(int), stream, ids_a, ids_b, T, k, agree);
    return d2s_check_launch();
}

int d2s_act_grad(const float* g, const float* z, float* out, long n, int kind, hipStream_t stream) {
    if (!g || !z || !out || n <= 0 || kind < 0 || kind > 1) return D2S_ERR_ARG;
    hipLaunchKernelGGL(act_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, z, out, n, kind);
    return d2s_check_launch();
}

int d2s_adamw_chunk_elems() { return CH; }

// params / grads / exp_avg / exp_avg_sq: flat arenas of n_chunks * 1024 floats; desc: n_chunks x {float lr, float wd,
// int active, int pad} in device memory.  chunk_steps: n_chunks ints in device memory, the per-tensor update counters of
// torch.optim.AdamW (state['step']), advanced by this call for the active chunks; null: `step` >= 1 is used for every chunk.
int d2s_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const void* desc, int n_chunks, float beta1,
                   float beta2, float eps, int step, float grad_scale, int* chunk_steps, hipStream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !desc || n_chunks <= 0 || (!chunk_steps && step < 1)) return D2S_ERR_ARG;
    const double bc1 = chunk_steps ? 1.0 : 1.0 - pow((double)beta1, (double)step);
    const double bc2 = chunk_steps ? 1.0 : 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq,
                       static_cast<const ChunkDesc*>(desc), beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale, chunk_steps);
    return d2s_check_launch();
}

}  // extern "C"
