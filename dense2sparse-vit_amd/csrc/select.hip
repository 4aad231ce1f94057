// Token scoring tail, hard top-k selection, gather/pack of the surviving tokens and its backward scatter.
// Reference: vit_models/dynamic_vit.py:540-551 (split / token-mean / concat, softmax over tokens),
// :852-865 (argsort descending, split at k, sort ids ascending), :907-912 (torch.gather of [0, kept+1]).
// All of it is HBM- or latency-bound integer / byte work: coalesced float4 rows, ids staged once per
// workgroup in LDS, wave ballots for the ordered compaction.  No atomics anywhere, results are deterministic.
#include "d2s_common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------------------
// softmax over the token axis: probs[r][t] = exp(s - max) * (1 / sum)   (same operation order as ATen's CPU
// vec_softmax so that near-ties resolve like the reference wherever fp32 allows)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, float* __restrict__ p, int T) {
    __shared__ float red[4];
    __shared__ float bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sr = s + (long)blockIdx.x * T;
    float* pr = p + (long)blockIdx.x * T;
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) m = fmaxf(m, sr[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bc;
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float e = expf(sr[t] - m);
        pr[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) bc = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    __syncthreads();
    const float inv = bc;
    for (int t = tid; t < T; t += 256) pr[t] = pr[t] * inv;
}

// ---------------------------------------------------------------------------------------------------------
// hard top-k: rank by counting (value descending, equal values lowest index first), ordered compaction.
// One workgroup per image, scores staged in LDS, emits both id lists already sorted ascending.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_topk_kernel(const float* __restrict__ probs, int T, int k,
                                                          long long* __restrict__ kept, long long* __restrict__ dropped) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [T] scores, then [T] keep flags (as int)
    float* ps = sh;
    int* flag = reinterpret_cast<int*>(sh + T);
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pr = probs + (long)blockIdx.x * T;
    for (int t = tid; t < T; t += 256) ps[t] = pr[t];
    __syncthreads();
    for (int i = tid; i < T; i += 256) {
        const float v = ps[i];
        int cnt = 0;
        for (int j = 0; j < T; ++j) {
            const float u = ps[j];
            cnt += (u > v) || (u == v && j < i);
        }
        flag[i] = cnt < k;
    }
    __syncthreads();
    long long* ko = kept + (long)blockIdx.x * k;
    long long* dr = dropped ? dropped + (long)blockIdx.x * (T - k) : nullptr;
    int base = 0;  // number of kept ids among indices below the current 256-chunk
    for (int c0 = 0; c0 < T; c0 += 256) {
        const int i = c0 + tid;
        const int f = (i < T) ? flag[i] : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        const int pos = base + woff + before;
        if (i < T) {
            if (f) { if (pos < k) ko[pos] = i; }
            else if (dr) dr[i - pos] = i;
        }
        base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// gather / pack:   out[b,0] = x[b,0];  out[b,1+j] = x[b,1+ids[b,j]]
// One wave per output row, 4 rows per workgroup: the row's source index is one wave-uniform load, then every lane issues all of
// its 16-byte loads before the first store.  No LDS, no barrier, no integer division per element.  (Measured against the
// earlier slab-per-workgroup form with LDS-staged indices: 9.9 vs 11.4 us at B=128, k=98, D=384 - tools/gather_bench.py.)
// ---------------------------------------------------------------------------------------------------------
constexpr int GROWS = 16;   // rows per workgroup of the scatter kernel below
__global__ __launch_bounds__(256) void gather_pack_kernel(const float* __restrict__ x, const long long* __restrict__ ids,
                                                                  float* __restrict__ out, int n, int k, int D, long rows_total) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // packed row index over the whole batch
    if (row >= rows_total) return;
    const int b = (int)(row / (k + 1)), r = (int)(row - (long)b * (k + 1));
    const int srow = r == 0 ? 0 : 1 + (int)ids[(long)b * k + (r - 1)];
    const int nvec = D >> 2;
    const f32x4* xs = reinterpret_cast<const f32x4*>(x + ((long)b * n + srow) * D);
    f32x4* od = reinterpret_cast<f32x4*>(out + row * D);
    f32x4 v[4];
    for (int c0 = 0; c0 < nvec; c0 += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u * 64 + lane < nvec) v[u] = xs[c0 + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u * 64 + lane < nvec) od[c0 + u * 64 + lane] = v[u];
    }
}

// (A wave-per-row scatter that finds its packed row by a ballot search over the kept ids measured the same, 12.8 vs 13.0 us.)
// backward: dx[b] = 0 except dx[b,0] (+)= g[b,0], dx[b,1+ids[b,j]] = g[b,1+j].  Every dx row is written exactly
// once (zeros for dropped tokens), so no atomics and no separate memset.  Inverse map built in LDS per workgroup.
__global__ __launch_bounds__(256) void scatter_unpack_kernel(const float* __restrict__ g, const long long* __restrict__ ids,
                                                             float* __restrict__ dx, int n, int k, int D) {
    extern __shared__ int inv[];  // [n] : packed row feeding dx row, or -1
    const int b = blockIdx.y, r0 = blockIdx.x * GROWS, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) inv[i] = i == 0 ? 0 : -1;
    __syncthreads();
    for (int j = tid; j < k; j += 256) {
        const int t = (int)ids[(long)b * k + j];
        if (t >= 0 && t < n - 1) inv[1 + t] = 1 + j;
    }
    __syncthreads();
    const int nrows = min(GROWS, n - r0);
    const int nvec = D >> 2;
    const int total = nrows * nvec;
    const f32x4* gb = reinterpret_cast<const f32x4*>(g + (long)b * (k + 1) * D);
    f32x4* ob = reinterpret_cast<f32x4*>(dx + ((long)b * n + r0) * D);
    for (int base = 0; base < total; base += 256 * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + u * 256 + tid;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < total) {
                const int r = e / nvec, c = e - r * nvec;
                const int s = inv[r0 + r];
                if (s >= 0) v[u] = gb[(long)s * nvec + c];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + u * 256 + tid;
            if (e < total) ob[e] = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// predictor split / token-mean / concat (dynamic_vit.py:540-544).  out[b,t,c] = x[b,t,c] for c < C/2 and
// mean_t x[b,t,c] for c >= C/2.  Its backward is the same operator applied to the gradient; `mask` (optional)
// multiplies the result by (mask > 0), which folds the preceding ReLU's backward into the same pass.
// grid: (column chunks of 64 over the two halves, B)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void half_mean_concat_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                               float* __restrict__ out, int T, int C) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = C >> 1;
    const int chunks = (half + 63) >> 6;
    const bool second = (int)blockIdx.x >= chunks;
    const int c = (second ? half : 0) + ((int)blockIdx.x - (second ? chunks : 0)) * 64 + lane;
    const bool ok = c < (second ? C : half);
    const long base = (long)blockIdx.y * T * C;
    if (!second) {
        if (ok)
            for (int t = wave; t < T; t += 4) {
                float v = x[base + (long)t * C + c];
                if (mask) v = mask[base + (long)t * C + c] > 0.f ? v : 0.f;
                out[base + (long)t * C + c] = v;
            }
        return;
    }
    float s = 0.f;
    if (ok)
        for (int t = wave; t < T; t += 4) s += x[base + (long)t * C + c];
    red[wave][lane] = s;
    __syncthreads();
    const float mean = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)T;
    if (ok)
        for (int t = wave; t < T; t += 4) {
            float v = mean;
            if (mask) v = mask[base + (long)t * C + c] > 0.f ? v : 0.f;
            out[base + (long)t * C + c] = v;
        }
}

// 16-byte form (C a multiple of 8): a lane owns 4 consecutive columns, a wave instruction moves 1 KB of a row (the scalar form
// below moved 256-byte segments: 2.9 TB/s on the 25088 x 1536 predictor activation), the four waves of a workgroup interleave over the tokens.
__global__ __launch_bounds__(256) void half_mean_concat_vec_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                                   float* __restrict__ out, int T, int C) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = C >> 1;
    const int chunks = (half + 255) >> 8;
    const bool second = (int)blockIdx.x >= chunks;
    const int c = (second ? half : 0) + ((int)blockIdx.x - (second ? chunks : 0)) * 256 + lane * 4;
    const bool ok = c < (second ? C : half);
    const long base = (long)blockIdx.y * T * C + c;
    if (!second) {
        if (ok)
#pragma unroll 4
            for (int t = wave; t < T; t += 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(x + base + (long)t * C);
                if (mask) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(mask + base + (long)t * C);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + base + (long)t * C) = v;
            }
        return;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok)
#pragma unroll 8
        for (int t = wave; t < T; t += 4) s += *reinterpret_cast<const f32x4*>(x + base + (long)t * C);
    red[wave][lane] = s;
    __syncthreads();
    f32x4 mean;
#pragma unroll
    for (int j = 0; j < 4; ++j) mean[j] = ((red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j])) / (float)T;
    if (ok)
#pragma unroll 4
        for (int t = wave; t < T; t += 4) {
            f32x4 v = mean;
            if (mask) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(mask + base + (long)t * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
            }
            *reinterpret_cast<f32x4*>(out + base + (long)t * C) = v;
        }
}

// ---------------------------------------------------------------------------------------------------------
// patch extraction (im2col of the stride-16 conv, dynamic_vit.py:298,305): image [B,Cin,H,W] -> [B*T, Cin*P*P]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, float* __restrict__ col, int Cin,
                                                           int H, int W, int P, long total_vec) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;  // float4 index in image order
    if (e >= total_vec) return;
    const int wv = W >> 2;
    const int x4 = (int)(e % wv);
    long r = e / wv;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % Cin);
    const long b = r / Cin;
    const int x = x4 * 4;
    const int ty = y / P, py = y - ty * P, tx = x / P, px = x - tx * P;
    const int Tw = W / P, T = (H / P) * Tw, K = Cin * P * P;
    const f32x4 v = reinterpret_cast<const f32x4*>(img)[e];
    *reinterpret_cast<f32x4*>(col + ((long)b * T + ty * Tw + tx) * K + c * P * P + py * P + px) = v;
}

// tokens[b,0,:] = cls + pos[0]  (dynamic_vit.py:820-823)
__global__ __launch_bounds__(256) void fill_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ tokens, int n, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < D) tokens[(long)blockIdx.y * n * D + d] = cls[d] + pos[d];
}

// out[r][d] (+)= sum_b g[b][r][d]   (pos_embed / cls_token gradients)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int B,
                                                        long per_image, long count, long image_stride, int accumulate) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[(long)b * image_stride + e];
    out[e] = accumulate ? out[e] + s : s;
    (void)per_image;
}

// out[b,0,:] = cls + pos[0];  out[b,1+t,:] = tok[b,t,:] + pos[1+t,:]   (t2t_vit.py:160-162: cat CLS, add pos_embed)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, float* __restrict__ out, int T, int D,
                                                              long total_vec) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;   // float4 index over [B, T+1, D]
    if (e >= total_vec) return;
    const int nvec = D >> 2;
    const int c = (int)(e % nvec);
    const long r = e / nvec;
    const int t = (int)(r % (T + 1));
    const long b = r / (T + 1);
    const f32x4 pv = reinterpret_cast<const f32x4*>(pos)[(long)t * nvec + c];
    const f32x4 sv = t == 0 ? reinterpret_cast<const f32x4*>(cls)[c]
                            : reinterpret_cast<const f32x4*>(tok)[((long)b * T + (t - 1)) * nvec + c];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = sv[j] + pv[j];
    reinterpret_cast<f32x4*>(out)[e] = o;
}

// generic row copy through a row map (strip the CLS rows of a gradient buffer, etc.)
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, long rows_per_group, long group_stride,
                                                        long row_stride, long offset, float* __restrict__ dst,
                                                        long d_rows_per_group, long d_group_stride, long d_row_stride,
                                                        long d_offset, long rows, int D, int accumulate) {
    const int nvec = D >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * nvec) return;
    const long r = e / nvec;
    const int c = (int)(e - r * nvec);
    const long g = r / rows_per_group, t = r - g * rows_per_group;
    const float* s = src + g * group_stride + offset + t * row_stride;
    const long dg = r / d_rows_per_group, dt = r - dg * d_rows_per_group;
    float* d = dst + dg * d_group_stride + d_offset + dt * d_row_stride;
    f32x4 v = reinterpret_cast<const f32x4*>(s)[c];
    if (accumulate) {
        const f32x4 o = reinterpret_cast<const f32x4*>(d)[c];
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    reinterpret_cast<f32x4*>(d)[c] = v;
}

}  // namespace

extern "C" {

int d2s_softmax_rows(const float* scores, float* probs, int rows, int T, hipStream_t stream) {
    if (!scores || !probs || rows <= 0 || T <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, scores, probs, T);
    return d2s_check_launch();
}

// probs [B,T] fp32 -> kept [B,k] int64 ascending, dropped [B,T-k] int64 ascending (may be null).  k is clamped to T.
int d2s_select_topk(const float* probs, int B, int T, int k, long long* kept, long long* dropped, hipStream_t stream) {
    if (!probs || (!kept && k > 0) || B <= 0 || T <= 0 || k < 0 || k > T || T > 16384) return D2S_ERR_ARG;
    if (k == 0 && !dropped) return D2S_OK;
    hipLaunchKernelGGL(select_topk_kernel, dim3(B), dim3(256), (size_t)2 * T * sizeof(float), stream, probs, T, k, kept,
                       (T - k) > 0 ? dropped : nullptr);
    return d2s_check_launch();
}

// x [B,n,D] -> out [B,k+1,D]; ids [B,k] int64, stage-relative (0-based over the n-1 non-CLS tokens)
int d2s_gather_pack_fwd(const float* x, const long long* ids, float* out, int B, int n, int k, int D, hipStream_t stream) {
    if (!x || (!ids && k > 0) || !out || B <= 0 || n <= 0 || k < 0 || k > n - 1 || D <= 0 || (D & 3)) return D2S_ERR_ARG;
    const long rows = (long)B * (k + 1);
    hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ids, out, n, k, D, rows);
    return d2s_check_launch();
}

// g [B,k+1,D] -> dx [B,n,D] (every element written)
int d2s_scatter_unpack_bwd(const float* g, const long long* ids, float* dx, int B, int n, int k, int D, hipStream_t stream) {
    if (!g || (!ids && k > 0) || !dx || B <= 0 || n <= 0 || k < 0 || k > n - 1 || D <= 0 || (D & 3)) return D2S_ERR_ARG;
    hipLaunchKernelGGL(scatter_unpack_kernel, dim3((n + GROWS - 1) / GROWS, B), dim3(256), (size_t)n * sizeof(int), stream, g,
                       ids, dx, n, k, D);
    return d2s_check_launch();
}

int d2s_half_mean_concat(const float* x, const float* relu_mask_src, float* out, int B, int T, int C, hipStream_t stream) {
    if (!x || !out || B <= 0 || T <= 0 || C <= 0 || (C & 1)) return D2S_ERR_ARG;
    const bool vec = (C % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(relu_mask_src)) & 15) == 0;
    if (vec) {
        const int chunks = ((C >> 1) + 255) >> 8;
        hipLaunchKernelGGL(half_mean_concat_vec_kernel, dim3(2 * chunks, B), dim3(256), 0, stream, x, relu_mask_src, out, T, C);
        return d2s_check_launch();
    }
    const int chunks = ((C >> 1) + 63) >> 6;
    hipLaunchKernelGGL(half_mean_concat_kernel, dim3(2 * chunks, B), dim3(256), 0, stream, x, relu_mask_src, out, T, C);
    return d2s_check_launch();
}

int d2s_im2col_patch(const float* img, float* col, int B, int Cin, int H, int W, int P, hipStream_t stream) {
    if (!img || !col || B <= 0 || Cin <= 0 || P <= 0 || (P & 3) || H % P || W % P) return D2S_ERR_ARG;
    const long total_vec = (long)B * Cin * H * (W >> 2);
    hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)((total_vec + 255) / 256)), dim3(256), 0, stream, img, col, Cin, H, W,
                       P, total_vec);
    return d2s_check_launch();
}

int d2s_fill_cls(const float* cls, const float* pos, float* tokens, int B, int n, int D, hipStream_t stream) {
    if (!cls || !pos || !tokens || B <= 0 || n <= 0 || D <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(fill_cls_kernel, dim3((D + 255) / 256, B), dim3(256), 0, stream, cls, pos, tokens, n, D);
    return d2s_check_launch();
}

// out[e] (+)= sum_b g[b*image_stride + e], e in [0,count)
int d2s_batch_sum(const float* g, float* out, int B, long count, long image_stride, int accumulate, hipStream_t stream) {
    if (!g || !out || B <= 0 || count <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, g, out, B, count, count,
                       image_stride, accumulate);
    return d2s_check_launch();
}

int d2s_assemble_tokens(const float* tok, const float* cls, const float* pos, float* out, int B, int T, int D, hipStream_t stream) {
    if (!tok || !cls || !pos || !out || B <= 0 || T <= 0 || D <= 0 || (D & 3)) return D2S_ERR_ARG;
    const long total_vec = (long)B * (T + 1) * (D >> 2);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((total_vec + 255) / 256)), dim3(256), 0, stream, tok, cls, pos, out, T, D,
                       total_vec);
    return d2s_check_launch();
}

// row r of src (through the source row map) -> row r of dst (through the destination row map)
int d2s_copy_rows(const float* src, long rows_per_group, long group_stride, long row_stride, long offset, float* dst,
                  long d_rows_per_group, long d_group_stride, long d_row_stride, long d_offset, long rows, int D,
                  int accumulate, hipStream_t stream) {
    if (!src || !dst || rows <= 0 || D <= 0 || (D & 3) || rows_per_group <= 0 || d_rows_per_group <= 0) return D2S_ERR_ARG;
    if ((group_stride | row_stride | offset | d_group_stride | d_row_stride | d_offset) & 3) return D2S_ERR_ARG;
    const long total = rows * (D >> 2);
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, rows_per_group,
                       group_stride, row_stride, offset, dst, d_rows_per_group, d_group_stride, d_row_stride, d_offset, rows, D,
                       accumulate);
    return d2s_check_launch();
}

}  // extern "C"
