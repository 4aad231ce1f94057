// Dynamic keep ratio (--patch-score-threshold): cumulative-score threshold selection, the ragged pack of the
// surviving tokens for inference, and the mask bookkeeping around them.
// Reference: vit_models/dynamic_vit.py:880-894 (training: ascending sort of the keep probabilities, cumulative sum, keep where the
// running sum exceeds the threshold, scatter back to token order), :935-949 (inference: the same selection, then only the kept
// tokens go on - one length per image), visualizations.py:18-26 (kept / dropped id lists -> 0/1 mask in patch order).
// Byte / integer work, one workgroup per image; nothing here is shaped for the matrix cores.
#include "d2s_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// threshold selection.  rank_i = #{j : p_j < p_i or (p_j == p_i and j < i)} (ascending, equal values lowest index first = a stable
// ascending sort); sorted[rank_i] = p_i; the cumulative sum runs SEQUENTIALLY in fp32 in sorted order (the summation order of
// torch.cumsum on the CPU, which decides tokens whose running sum lands within rounding of the threshold); keep_i = cum[rank_i] > th.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_threshold_kernel(const float* __restrict__ probs, int T, float threshold,
                                                               float* __restrict__ mask, long ld, int lead, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [T] values, [T] sorted values -> cumulative sums, [T] ranks
    float* ps = sh;
    float* srt = sh + T;
    int* rank = reinterpret_cast<int*>(sh + 2 * T);
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
            cnt += (u < v) || (u == v && j < i);
        }
        rank[i] = cnt;
        srt[cnt] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float run = 0.f;
        for (int r = 0; r < T; ++r) {
            run += srt[r];
            srt[r] = run;
        }
    }
    __syncthreads();
    int kept = 0;
    for (int i = tid; i < T; i += 256) {
        const int f = srt[rank[i]] > threshold;
        mask[(long)blockIdx.x * ld + lead + i] = f ? 1.f : 0.f;
        kept += f;
    }
    if (tid < lead) mask[(long)blockIdx.x * ld + tid] = 1.f;       // the CLS slot(s) of an attention policy row: always kept (:892-893)
    kept = wave_sum_i(kept);
    if (lane == 0) wave_tot[wave] = kept;
    __syncthreads();
    if (tid == 0 && counts) counts[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// cu[0] = 0, cu[b+1] = cu[b] + counts[b] + extra   (extra = 1: the CLS token that every image keeps)
__global__ __launch_bounds__(256) void ragged_offsets_kernel(const int* __restrict__ counts, int B, int extra, int* __restrict__ cu) {
    if (threadIdx.x == 0) {
        int run = 0;
        cu[0] = 0;
        for (int b = 0; b < B; ++b) {
            run += counts[b] + extra;
            cu[b + 1] = run;
        }
    }
}

// Ragged pack: image b contributes its CLS row and the rows whose mask is 1, in token order, to out[cu[b] .. cu[b+1]).
// One workgroup per image: positions from ballot prefix sums over the mask (no atomics), then one wave per output row copies it
// with 16-byte accesses.  row_src (optional, [total]) receives the source token index of every packed row.
__global__ __launch_bounds__(256) void ragged_pack_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          const int* __restrict__ cu, float* __restrict__ out, int* __restrict__ row_src,
                                                          int n, int D) {
    extern __shared__ int src[];       // [n] source token of packed row j of this image
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* mb = mask + (long)b * (n - 1);
    if (tid == 0) src[0] = 0;
    int base = 1;
    for (int c0 = 0; c0 < n - 1; c0 += 256) {
        const int i = c0 + tid;
        const int f = (i < n - 1) ? (mb[i] != 0.f) : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        if (f) src[base + woff + before] = i + 1;
        base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int o0 = cu[b], cnt = cu[b + 1] - o0;     // == base when the offsets were built from this mask
    const float* xb = x + (long)b * n * D;
    const int nv = D >> 2;
    for (int j = wave; j < cnt; j += 4) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(xb + (long)src[j] * D);
        f32x4* d4 = reinterpret_cast<f32x4*>(out + (long)(o0 + j) * D);
        for (int c = lane; c < nv; c += 64) d4[c] = s4[c];
        if (row_src && lane == 0) row_src[o0 + j] = src[j];
    }
}

// w[r] = mask[r] / sum(mask): turns "mean over the rows a mask keeps" into a weighted row sum (the token-distillation term restricted
// to the kept tokens, the build's fix for the undefined `C` / flat-mask indexing at losses.py:216-218).  Single workgroup.
__global__ __launch_bounds__(1024) void mask_row_weights_kernel(const float* __restrict__ mask, long rows, float* __restrict__ w) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (long r = tid; r < rows; r += 1024) s += mask[r];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float inv = tot > 0.f ? 1.0f / tot : 0.f;
    for (long r = tid; r < rows; r += 1024) w[r] = mask[r] * inv;
}

// agree[b] = #{t : a[b,t] == b_[b,t]}   (mask accuracy between two dense 0/1 masks)
__global__ __launch_bounds__(256) void dense_mask_agreement_kernel(const float* __restrict__ a, const float* __restrict__ b_, int T,
                                                                   float* __restrict__ agree) {
    __shared__ int red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 0;
    for (int t = tid; t < T; t += 256) c += a[(long)blockIdx.x * T + t] == b_[(long)blockIdx.x * T + t];
    c = wave_sum_i(c);
    if (lane == 0) red[wave] = c;
    __syncthreads();
    if (tid == 0) agree[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// visualizations.py:18-26: the kept / dropped id lists of one stage as a 0/1 mask in token order (kept -> 1).
__global__ __launch_bounds__(256) void patch_keep_mask_kernel(const long long* __restrict__ kept, int k, int N, long long* __restrict__ mask) {
    long long* mb = mask + (long)blockIdx.x * N;
    for (int t = threadIdx.x; t < N; t += 256) mb[t] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        const long long id = kept[(long)blockIdx.x * k + j];
        if (id >= 0 && id < N) mb[id] = 1;
    }
}

// stage-relative ids -> ids in the coordinates of the previous stage's input: out[b,j] = prev[b, rel[b,j]]
__global__ __launch_bounds__(256) void compose_ids_kernel(const long long* __restrict__ prev, int kp, const long long* __restrict__ rel, int k,
                                                          long long* __restrict__ out) {
    for (int j = threadIdx.x; j < k; j += 256) {
        const long long r = rel[(long)blockIdx.x * k + j];
        out[(long)blockIdx.x * k + j] = (r >= 0 && r < kp) ? prev[(long)blockIdx.x * kp + r] : -1;
    }
}

// dst[r] = src[idx[r]] (rows of D floats): the CLS rows of a ragged packed batch (idx = cu_seqlens) for the classifier head
__global__ __launch_bounds__(256) void gather_rows_i32_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                                              int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (long)idx[r] * D);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + (long)r * D);
    for (int c = lane; c < (D >> 2); c += 64) d4[c] = s4[c];
}

}  // namespace

extern "C" {

// probs [B,T] -> mask (1.0 = kept): token i of image b at mask[b * ld + lead + i]; the `lead` leading entries of each row are set to
// 1 (lead = 1, ld = T + 1 gives the attention policy row [CLS, tokens] directly); counts [B] (may be null).  T <= 8192.
int d2s_select_threshold(const float* probs, int B, int T, float threshold, float* mask, long ld, int lead, int* counts,
                         hipStream_t stream) {
    if (!probs || !mask || B <= 0 || T <= 0 || T > 8192 || lead < 0 || lead > 256 || ld < T + lead) return D2S_ERR_ARG;
    hipLaunchKernelGGL(select_threshold_kernel, dim3(B), dim3(256), (size_t)3 * T * sizeof(float), stream, probs, T, threshold, mask, ld,
                       lead, counts);
    return d2s_check_launch();
}

int d2s_ragged_offsets(const int* counts, int B, int extra, int* cu_seqlens, hipStream_t stream) {
    if (!counts || !cu_seqlens || B <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(ragged_offsets_kernel, dim3(1), dim3(64), 0, stream, counts, B, extra, cu_seqlens);
    return d2s_check_launch();
}

// x [B,n,D] (row 0 = CLS), mask [B,n-1], cu_seqlens [B+1] from d2s_ragged_offsets(counts, B, 1) -> out [cu[B], D].
int d2s_ragged_pack(const float* x, const float* mask, const int* cu_seqlens, float* out, int* row_src, int B, int n, int D,
                    hipStream_t stream) {
    if (!x || !mask || !cu_seqlens || !out || B <= 0 || n <= 1 || n > 8192 || D <= 0 || (D & 3)) return D2S_ERR_ARG;
    hipLaunchKernelGGL(ragged_pack_kernel, dim3(B), dim3(256), (size_t)n * sizeof(int), stream, x, mask, cu_seqlens, out, row_src, n, D);
    return d2s_check_launch();
}

int d2s_gather_rows_i32(const float* src, const int* idx, float* dst, int rows, int D, hipStream_t stream) {
    if (!src || !idx || !dst || rows <= 0 || D <= 0 || (D & 3)) return D2S_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_i32_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, src, idx, dst, rows, D);
    return d2s_check_launch();
}

int d2s_mask_row_weights(const float* mask, long rows, float* weights, hipStream_t stream) {
    if (!mask || !weights || rows <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(mask_row_weights_kernel, dim3(1), dim3(1024), 0, stream, mask, rows, weights);
    return d2s_check_launch();
}

int d2s_dense_mask_agreement(const float* mask_a, const float* mask_b, int B, int T, float* agree, hipStream_t stream) {
    if (!mask_a || !mask_b || !agree || B <= 0 || T <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(dense_mask_agreement_kernel, dim3(B), dim3(256), 0, stream, mask_a, mask_b, T, agree);
    return d2s_check_launch();
}

int d2s_patch_keep_mask(const long long* kept, int B, int k, int N, long long* mask, hipStream_t stream) {
    if (!mask || B <= 0 || N <= 0 || k < 0 || k > N || (k > 0 && !kept)) return D2S_ERR_ARG;
    hipLaunchKernelGGL(patch_keep_mask_kernel, dim3(B), dim3(256), 0, stream, kept, k, N, mask);
    return d2s_check_launch();
}

int d2s_compose_ids(const long long* prev, int kp, const long long* rel, int k, long long* out, int B, hipStream_t stream) {
    if (!prev || !rel || !out || B <= 0 || kp <= 0 || k <= 0) return D2S_ERR_ARG;
    hipLaunchKernelGGL(compose_ids_kernel, dim3(B), dim3(256), 0, stream, prev, kp, rel, k, out);
    return d2s_check_launch();
}

}  // extern "C"
