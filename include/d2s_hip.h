/* d2s_hip.h - C ABI of libd2s_hip.so: the MI355X (gfx950) kernels of the dense-to-sparse ViT training path.
 *
 * The reference (marc345/Dense2Sparse-ViT) is 100 % Python on top of ATen; it has no native boundary of its own.  Each
 * entry point below replaces the ATen call(s) made at the cited reference lines (paths relative to the reference
 * repository root) and is what a ctypes / cffi binding on the reference side would bind (INTEGRATION.md shows it).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors kept alive by the Python wrapper);
 *   - fp32 data, int64 token ids (`long long`), row-major, innermost dimension dense;
 *   - every launching function returns 0 on success, D2S_ERR_* (< 0) otherwise; nothing throws, nothing allocates:
 *     scratch is passed in (`workspace`, sized by the matching *_workspace_bytes query);
 *   - work is enqueued asynchronously on `stream`; calls on different streams need different workspaces;
 *   - "row map" = (rows_per_group, group_stride, row_stride, offset): element offset of logical row r is
 *       (r / rows_per_group) * group_stride + offset + (r % rows_per_group) * row_stride
 *     which lets a kernel read x[:, 1:] of a [B, n, D] buffer in place (vit_models/dynamic_vit.py:855).
 */
#ifndef D2S_HIP_H
#define D2S_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* d2s_stream_t; /* == hipStream_t */

#define D2S_OK 0
#define D2S_ERR_ARG (-1)
#define D2S_ERR_WORKSPACE (-2)
#define D2S_ERR_LAUNCH (-3)

/* ---- GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32) -----------------------------------------------
 * Replaces nn.Linear / Conv2d forward and their autograd backward: vit_models/dynamic_vit.py:169-175 (Mlp),
 * :218,:231 (qkv, proj), :298-305 (patch conv as GEMM), :491-531 (predictor layers), :1006 (head).
 * layout 0 "NT": C[M,N] = A[M,K] * B[N,K]^T (forward); 1 "NN": C[M,N] = A[M,K] * B[K,N] (dgrad);
 * layout 2 "TN": C[M,N] = A[K,M]^T * B[K,N] (wgrad; split-K slabs in `workspace`, deterministic combine).
 * epilogue: 0 none, 1 +bias[n], 2 relu(+bias), 3 gelu(+bias) with pre-activation copy to aux_out, 4 +bias+aux[m][n],
 *           5 *gelu'(aux[m][n]), 6 *(aux[m][n] > 0), 7 +bias+aux[m % aux_rows][n] (pos_embed add), 8 C += acc.
 * remap_rows_per_img/remap_skip: output row m goes to m + (m / rows_per_img + 1) * skip (leave room for CLS rows). */
/* GEMM arithmetic mode (NT / NN layouts in modes 1 and 2; the weight gradient follows in mode 2 only, its bias gradient stays an
 * exact fp32 column sum): 0 = exact fp32 MFMA; 1 = "bf16x3 split":
 * each fp32 operand is split into three bf16 pieces in registers and the 6 significant cross products run on the bf16 matrix
 * cores with fp32 accumulation (error ~2^-24 per product, fp32-class; 416 TFLOP/s effective peak); 2 = bf16 operands, fp32
 * accumulation (BASELINE config 5's bf16 regime).  Inputs and outputs stay fp32 in every mode.  `mode` is an argument of every
 * call and of the matching workspace query (no process-global state: calls in different modes may run concurrently on different
 * streams / threads, e.g. a bf16 teacher beside an fp32 student). */
size_t d2s_gemm_f32_workspace_bytes(int layout, int M, int N, int K, int mode);
int d2s_gemm_f32(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                 int epilogue, const float* bias, const float* aux, long ldaux, float* aux_out, int aux_rows,
                 int remap_rows_per_img, int remap_skip, int accumulate, int mode, void* workspace, size_t workspace_bytes,
                 d2s_stream_t stream);
/* bf16 data path of mode 2 (BASELINE config 5's bf16 regime; what torch.autocast does around F.linear, vit_models/dynamic_vit.py:169-175,
 * 218-231): the same GEMM with bf16 side channels.  a_bf16 (optional): the A operand already rounded to bf16, dense [M][K], K % 32 == 0 -
 * the call skips its conversion pass over A (A, if given, must hold the same values; it may be NULL).  c_bf16 (optional): a dense [M][N]
 * bf16 copy of the result, N % 32 == 0, written by the same epilogue - the a_bf16 of the next GEMM (C may then be NULL).  Producers of
 * a_bf16 other than a GEMM: d2s_layernorm_fwd_bf16out, d2s_attn_fwd_bf16_bf16out.  NT / NN layouts; workspace as d2s_gemm_f32 in mode 2.
 * Two more epilogue codes, this entry only: 9 = 3 with aux_out pointing at bf16 [M][ldc] (fc1's pre-activation kept in bf16 for the
 * backward, as autocast keeps it), 10 = 5 with aux pointing at bf16 [M][ldaux] (the input gradient of fc2 reading it back). */
int d2s_gemm_f32_bf16io(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                        int epilogue, const float* bias, const float* aux, long ldaux, float* aux_out, const void* a_bf16, const void* b_bf16,
                        void* c_bf16, void* workspace, size_t workspace_bytes, d2s_stream_t stream);
/* dst[i] = bf16(src[i]): the bf16 form of a weight - or of a whole parameter arena in one launch - for b_bf16 above (b_bf16 is always
 * [N][K] k-contiguous: the weight itself for layout 0, W^T for layout 1; made once per optimiser step, once ever for a frozen model;
 * B may then be NULL) */
int d2s_convert_bf16(const float* src, void* dst, long n, d2s_stream_t stream);
/* nn.Linear parameter gradients in one pass over dy (autograd of F.linear at vit_models/dynamic_vit.py:169-175,218,231,491-531):
 * dW[n_out,n_in] (+)= dy[tokens,n_out]^T x[tokens,n_in];  db[n_out] (+)= column sums of dy (db may be NULL).  Exact fp32 MFMA,
 * deterministic split-K over the token rows; the bias gradient is folded out of the dy tiles the GEMM streams anyway. */
size_t d2s_linear_wgrad_workspace_bytes(int tokens, int n_out, int n_in, int mode);
int d2s_linear_wgrad_f32(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, float* db, int tokens,
                         int n_out, int n_in, int accumulate, int mode, void* workspace, size_t workspace_bytes, d2s_stream_t stream);
/* mode 2, bf16 data path: the same with the layer input available in bf16 only (x_bf16 [tokens][n_in], what the forward saved instead of
 * the fp32 activation).  dy is fp32 - its exact column sums are the bias gradient - or, where the kernel that produced the gradient wrote
 * bf16 only, dy_bf16 [tokens][n_out] (dy may then be NULL; the bias gradient is the fp32 sum of the bf16 values, as under torch.autocast).
 * Both may be given (dy_bf16 = dy rounded to bf16): with tokens % 64 == 0 and an output of 256x256 tiles the matrix kernel then reads
 * dy_bf16 and x_bf16 token-major as they lie (no transposing pass), the bias gradient stays the exact column sum of dy, and the result
 * is bit-identical to the dy-only call.  Workspace: d2s_linear_wgrad_workspace_bytes(.., 2). */
int d2s_linear_wgrad_f32_bf16x(const float* dy, const void* dy_bf16, long lddy, const void* x_bf16, long ldx, float* dW, long lddw, float* db,
                               int tokens, int n_out, int n_in, int accumulate, void* workspace, size_t workspace_bytes, d2s_stream_t stream);
/* dst[C][R] = src[R][C]^T: a k-contiguous copy W^T of an nn.Linear weight, so that autograd's dx = dy W (vit_models/dynamic_vit.py:169-175
 * backward) can run in the NT layout - both operands k-contiguous - instead of NN. */
int d2s_transpose_f32(const float* src, float* dst, int R, int C, d2s_stream_t stream);
/* every weight of a parameter arena in one launch: tile_desc = n_tiles x {long src_off, long dst_off, int R, int C, int r0, int c0}
 * (offsets in floats from the two bases; one 64x64 tile per descriptor), refreshed once per optimiser step */
int d2s_transpose_batched_f32(const float* src_base, float* dst_base, const void* tile_desc, int n_tiles, d2s_stream_t stream);
/* out[n] (+)= sum_m X[m][n]: bias gradients where no weight gradient is wanted. */
size_t d2s_colsum_workspace_bytes(int M, int N);
int d2s_colsum_f32(const float* X, long ldx, int M, int N, float* out, int accumulate, void* workspace,
                   size_t workspace_bytes, d2s_stream_t stream);

/* ---- LayerNorm (vit_models/dynamic_vit.py:245,250,678,993 and the predictor's LayerNorms :491-531) -------------- */
int d2s_layernorm_fwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w,
                      const float* b, float* y, float* mean, float* rstd, long rows, int D, float eps, d2s_stream_t stream);
/* the same with a dense [rows, D] bf16 copy of the output for a following bf16-mode GEMM (y may be NULL when only that form is consumed) */
int d2s_layernorm_fwd_bf16out(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w,
                              const float* b, float* y, void* y_bf16, float* mean, float* rstd, long rows, int D, float eps,
                              d2s_stream_t stream);
size_t d2s_layernorm_bwd_workspace_bytes(long rows, int D);
/* dx[map(r)] = (add_src ? add_src[map(r)] : 0) + mask * dLN/dx; relu_mask folds a preceding ReLU's backward in. */
int d2s_layernorm_bwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy,
                      const float* w, const float* mean, const float* rstd, float* dx, const float* add_src, float* dweight,
                      float* dbias, int accumulate_wb, int relu_mask, long rows, int D, void* workspace,
                      size_t workspace_bytes, d2s_stream_t stream);

/* the same backward with a dense [rows, D] bf16 copy of dx (after the residual add): the a_bf16 of the input-gradient GEMM that follows */
int d2s_layernorm_bwd_bf16out(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy,
                              const float* w, const float* mean, const float* rstd, float* dx, void* dx_bf16, const float* add_src,
                              float* dweight, float* dbias, int accumulate_wb, int relu_mask, long rows, int D, void* workspace,
                              size_t workspace_bytes, d2s_stream_t stream);

/* ---- fused attention, head dim 64 (vit_models/dynamic_vit.py:218-236; CLS row returned at :234) ------------------
 * qkv: raw output of the qkv Linear, [B, n, 3, H, 64]; out/dout: [B, n, H*64]; lse, cls_row, delta_ws: [B, H, n]. */
int d2s_attn_fwd_f32(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale,
                     d2s_stream_t stream);
/* Same contract on the bf16 matrix cores (Q, K, V rounded to bf16; scores, softmax, accumulation fp32): the forward of the bf16
 * arithmetic mode.  The fp32 backward below works from its outputs unchanged. */
int d2s_attn_fwd_bf16(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale,
                      d2s_stream_t stream);
/* ... on the bf16 data path: qkv may be given in bf16 (qkv_is_bf16 != 0: the c_bf16 of the qkv GEMM, same layout - the values the
 * kernel rounds to itself, so results are identical), and a dense [B, n, H*64] bf16 copy of the output is written for the projection
 * GEMM (out may be NULL in forward-only passes) */
int d2s_attn_fwd_bf16_bf16out(const void* qkv, int qkv_is_bf16, float* out, void* out_bf16, float* lse, float* cls_row, int B, int n, int H,
                              float scale, d2s_stream_t stream);
int d2s_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta_ws,
                     int B, int n, int H, float scale, d2s_stream_t stream);
/* the two independent halves of d2s_attn_bwd_f32 (delta_ws [B,H,n] from d2s_attn_delta): dQ, and dK / dV; they write disjoint parts of
 * dqkv, so they may run concurrently on two streams */
int d2s_attn_bwd_dq_f32(const float* qkv, const float* dout, const float* lse, const float* delta_ws, float* dqkv, int B, int n, int H,
                        float scale, d2s_stream_t stream);
int d2s_attn_bwd_dkv_f32(const float* qkv, const float* dout, const float* lse, const float* delta_ws, float* dqkv, int B, int n, int H,
                         float scale, d2s_stream_t stream);
/* delta[b,h,i] = sum_d dout * out (row term of the softmax backward); the two backward entries call it themselves */
int d2s_attn_delta(const float* out, const float* dout, float* delta, int B, int n, int H, d2s_stream_t stream);
/* backward on the bf16 matrix cores (bf16 arithmetic mode), same contract as d2s_attn_bwd_f32 */
int d2s_attn_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta_ws,
                     int B, int n, int H, float scale, d2s_stream_t stream);
/* ... on the bf16 data path: qkv optionally in bf16 (as in the forward) and a bf16 copy of dqkv in the same [B,n,3,H,64] layout (the
 * a_bf16 of the qkv Linear's input-gradient GEMM and the dy_bf16 of its weight gradient; dqkv may be NULL when only that form is used) */
int d2s_attn_bwd_bf16_bf16out(const void* qkv, int qkv_is_bf16, const float* out, const float* dout, const float* lse, float* dqkv,
                              void* dqkv_bf16, float* delta_ws, int B, int n, int H, float scale, d2s_stream_t stream);

/* Attention.softmax_with_policy (vit_models/dynamic_vit.py:195-214) on materialised scores [B,H,N,N], policy [B,N];
 * the backward includes the path through the row maximum (the reference does not detach it). */
int d2s_softmax_policy_fwd(const float* attn, const float* policy, float* out, int B, int H, int N, float eps, d2s_stream_t stream);
int d2s_softmax_policy_bwd(const float* attn, const float* policy, const float* grad_out, float* grad_attn, int B, int H, int N,
                           float eps, d2s_stream_t stream);

/* ---- token scoring tail, selection, gather / scatter ------------------------------------------------------------- */
/* F.softmax(scores, dim=-1), vit_models/dynamic_vit.py:551 */
int d2s_softmax_rows(const float* scores, float* probs, int rows, int T, d2s_stream_t stream);
/* argsort(desc)[:k] / [k:] then sort ascending, vit_models/dynamic_vit.py:858-862; ties: lowest index first */
int d2s_select_topk(const float* probs, int B, int T, int k, long long* kept, long long* dropped, d2s_stream_t stream);
/* torch.gather(x, 1, [0, kept+1]), vit_models/dynamic_vit.py:907-912 (train) / :954-960 (eval), and its backward */
int d2s_gather_pack_fwd(const float* x, const long long* ids, float* out, int B, int n, int k, int D, d2s_stream_t stream);
int d2s_scatter_unpack_bwd(const float* g, const long long* ids, float* dx, int B, int n, int k, int D, d2s_stream_t stream);
/* split / token-mean / concat of the predictor, vit_models/dynamic_vit.py:540-544 (self-adjoint: also its backward) */
int d2s_half_mean_concat(const float* x, const float* relu_mask_src, float* out, int B, int T, int C, d2s_stream_t stream);

/* ---- patch embedding helpers (vit_models/dynamic_vit.py:298-306, :820-823) ---------------------------------------- */
int d2s_im2col_patch(const float* img, float* col, int B, int Cin, int H, int W, int P, d2s_stream_t stream);
int d2s_fill_cls(const float* cls, const float* pos, float* tokens, int B, int n, int D, d2s_stream_t stream);
int d2s_batch_sum(const float* g, float* out, int B, long count, long image_stride, int accumulate, d2s_stream_t stream);
int d2s_copy_rows(const float* src, long rows_per_group, long group_stride, long row_stride, long offset, float* dst,
                  long d_rows_per_group, long d_group_stride, long d_row_stride, long d_offset, long rows, int D,
                  int accumulate, d2s_stream_t stream);

/* ---- Tokens-to-Token front end (vit_models/t2t_vit.py:45-104, token_performer.py:31-54) ------------------------------
 * soft split = nn.Unfold(k, s, p)(x).transpose(1, 2); the source is addressed by strides (sb, sc, sy, sx) so that an NCHW
 * image and the re-structurised token tensor [B, H*W, C] (t2t_vit.py:90,97) are both read in place. */
/* t2t_vit.py:160-162: out[b,0] = cls + pos[0]; out[b,1+t] = tok[b,t] + pos[1+t] */
int d2s_assemble_tokens(const float* tok, const float* cls, const float* pos, float* out, int B, int T, int D, d2s_stream_t stream);
int d2s_unfold_fwd(const float* src, long sb, long sc, long sy, long sx, float* out, int B, int C, int H, int W, int k, int s,
                   int p, d2s_stream_t stream);
int d2s_unfold_bwd(const float* g, float* dsrc, long sb, long sc, long sy, long sx, int B, int C, int H, int W, int k, int s,
                   int p, d2s_stream_t stream);
/* FAVOR+ linear attention of Token_performer.single_attn (token_performer.py:45-50), emb 64, m 32: kqv rows [k|q|v]. */
size_t d2s_performer_workspace_bytes(int B, int T);
int d2s_performer_attn_fwd(const float* kqv, const float* w, float* y, float* kp, float* qp, float* A, float* ksum, float* D,
                           int B, int T, float eps, void* workspace, size_t workspace_bytes, d2s_stream_t stream);
int d2s_performer_attn_bwd(const float* kqv, const float* w, const float* y, const float* kp, const float* qp, const float* A,
                           const float* ksum, const float* D, const float* gy, const float* skip, float* dkqv, float* dnum,
                           float* dD, float* dqp, float* dkp, float* dA, float* dksum, int B, int T, float eps, void* workspace,
                           size_t workspace_bytes, d2s_stream_t stream);

/* ---- perturbed top-k (vit_models/peturbed_topk.py:16-80), noise injected ------------------------------------------ */
/* peturbed_topk.py:29 draws torch.normal on the host; this fills out[0..n) with standard normals of the counter-based stream `seed`
 * (Philox4x32-10 + Box-Muller; element i does not depend on the launch shape) so that production needs no host RNG / H2D copy */
int d2s_normal_noise(float* out, long n, unsigned long long seed, d2s_stream_t stream);
size_t d2s_perturbed_topk_workspace_bytes(int b, int k, int d);
int d2s_perturbed_topk_fwd(const float* x, const float* noise, float* indicators, int b, int nS, int d, int k, float sigma,
                           void* workspace, size_t workspace_bytes, d2s_stream_t stream);
int d2s_perturbed_topk_bwd(const float* x, const float* noise, const float* grad_out, float* grad_x, int b, int nS, int d, int k,
                           float sigma, d2s_stream_t stream);

/* ---- losses (losses.py) -------------------------------------------------------------------------------------------- */
/* losses.py:76-79: mean over layers, max over heads, drop CLS column, renormalise. cls_attn [B,L,H,n] -> [B,n-1] */
int d2s_teacher_target(const float* cls_attn, float* target, int B, int L, int H, int n, d2s_stream_t stream);
/* losses.py:84-90: out[b,j] = in[b,ids[b,j]] (/ row sum when normalize) */
int d2s_gather_renorm(const float* in, const long long* ids, float* out, int B, int T, int k, int normalize, d2s_stream_t stream);
/* per-row loss + d/ds: mode 0 KL(logsm(s)||logsm(t)) (losses.py:198-203,220-225), 1 KL with t as probabilities (:94-95),
 * 2 cross entropy with labels (:196), 3 sum of squared differences s - t (the mse mask loss, :61-73), 4 soft-target cross entropy
 * -sum t log_softmax(s) (mixup labels, :170-172).  t rows: row map plus optional t_ids[r] * t_row_stride (teacher tokens gathered
 * by the kept ids, losses.py:212). */
int d2s_kl_rows(const float* s, long s_rpg, long s_gs, long s_rs, long s_off, const float* t, long t_rpg, long t_gs, long t_rs,
                long t_off, const long long* t_ids, const long long* labels, float* loss_row, float* grad, long rows, int C,
                int mode, const float* row_weight, d2s_stream_t stream);   /* row_weight [rows] or NULL: scales each row's loss and gradient */
int d2s_sum_scalar(const float* v, long n, float scale, float* out, d2s_stream_t stream);
int d2s_scale_by_scalar(const float* x, const float* gscalar, float scale, float* y, long n, d2s_stream_t stream);
/* losses.py:96,121-164: number of positions on which two top-k masks (as id lists) agree */
int d2s_mask_agreement(const long long* ids_a, const long long* ids_b, int B, int T, int k, float* agree, d2s_stream_t stream);
int d2s_act_grad(const float* g, const float* z, float* out, long n, int kind, d2s_stream_t stream);

/* ---- dynamic keep ratio (--patch-score-threshold) -------------------------------------------------------
 * vit_models/dynamic_vit.py:880-891: val, idx = sort(pred_score); th = cumsum(val) > threshold; mask = scatter(idx, th).
 * probs [B,T] -> mask (1.0 = kept) with token i of image b at mask[b * ld + lead + i] and the `lead` first entries of each row set
 * to 1 (lead = 1, ld = T + 1: the attention policy row [CLS, tokens] of :892-893); counts [B] (may be NULL).  Stable ascending order,
 * sequential fp32 running sum (torch.cumsum's order on the CPU). */
int d2s_select_threshold(const float* probs, int B, int T, float threshold, float* mask, long ld, int lead, int* counts,
                         d2s_stream_t stream);
/* dst[r] = src[idx[r]] for rows of D floats (the CLS rows of a ragged packed batch, idx = cu_seqlens) */
int d2s_gather_rows_i32(const float* src, const int* idx, float* dst, int rows, int D, d2s_stream_t stream);
/* :935-949 (inference keeps only the selected tokens; one length per image): cu[0] = 0, cu[b+1] = cu[b] + counts[b] + extra */
int d2s_ragged_offsets(const int* counts, int B, int extra, int* cu_seqlens, d2s_stream_t stream);
/* x [B,n,D] (row 0 = CLS, always kept), mask [B,n-1] -> out [cu[B], D]; row_src [cu[B]] (optional) = source token of each row */
int d2s_ragged_pack(const float* x, const float* mask, const int* cu_seqlens, float* out, int* row_src, int B, int n, int D,
                    d2s_stream_t stream);
/* weights[r] = mask[r] / sum(mask): the token-distillation term over the kept tokens only (the build's fix for losses.py:216-218) */
int d2s_mask_row_weights(const float* mask, long rows, float* weights, d2s_stream_t stream);
/* agree[b] = number of positions where two dense 0/1 masks [B,T] agree (mask accuracy, losses.py:96 for threshold masks) */
int d2s_dense_mask_agreement(const float* mask_a, const float* mask_b, int B, int T, float* agree, d2s_stream_t stream);
/* visualizations.py:18-26: kept ids [B,k] of one stage -> int64 0/1 mask [B,N] in token order */
int d2s_patch_keep_mask(const long long* kept, int B, int k, int N, long long* mask, d2s_stream_t stream);
/* out[b,j] = prev[b, rel[b,j]]: stage-relative kept ids expressed in the previous stage's coordinates */
int d2s_compose_ids(const long long* prev, int kp, const long long* rel, int k, long long* out, int B, d2s_stream_t stream);
/* Attention.forward with policy != None (vit_models/dynamic_vit.py:216-236 + softmax_with_policy :195-214) as one fused pass;
 * policy [B,n] (1 = kept, entry 0 = CLS); lse [B,H,n] = m + log(l + eps) and cinv [B,H,n] = (eps/n)/(l + eps) feed the backward. */
int d2s_attn_policy_fwd_f32(const float* qkv, const float* policy, float* out, float* lse, float* cinv, float* cls_row, int B, int n,
                            int H, float scale, float eps, d2s_stream_t stream);
int d2s_attn_policy_bwd_f32(const float* qkv, const float* policy, const float* out, const float* dout, const float* lse,
                            const float* cinv, float* dqkv, float* delta_ws, int B, int n, int H, float scale, d2s_stream_t stream);
/* ragged packed attention forward (inference): qkv [total,3,H,64], image b = rows cu[b]..cu[b+1]; cls_row (optional) [H,total] */
int d2s_attn_varlen_fwd_f32(const float* qkv, const int* cu_seqlens, float* out, float* cls_row, int B, int total, int max_n, int H,
                            float scale, d2s_stream_t stream);

/* ---- BatchNorm1d over the token rows of [R, C] (the --predictor-bn variant: vit_models/dynamic_vit.py:350-367 BatchNormLayer) ------ */
size_t d2s_batchnorm_workspace_bytes(long R, int C);
/* training != 0: batch statistics (saved to mean / rstd; running estimates updated with `momentum`, unbiased variance, when given);
 * training == 0: the running estimates are used (and copied into mean / rstd for the backward). */
int d2s_batchnorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, float* running_mean,
                      float* running_var, long R, int C, float eps, float momentum, int training, void* workspace,
                      size_t workspace_bytes, d2s_stream_t stream);
/* dx always; dw / db (+)= when given; relu_mask != 0 zeroes dx where x <= 0 (a ReLU whose output is this layer's input). */
int d2s_batchnorm_bwd(const float* x, const float* dy, const float* w, const float* mean, const float* rstd, float* dx, float* dw,
                      float* db, int relu_mask, int accumulate, int training, long R, int C, void* workspace, size_t workspace_bytes,
                      d2s_stream_t stream);

/* ---- one transformer block per call (csrc/block.hip) ------------------------------------------------------------------------------
 * Block.forward (vit_models/dynamic_vit.py:263-283: x + proj(attn(LN1 x)), then + fc2(gelu(fc1(LN2 .))), with Attention :216-236 and Mlp
 * :169-175) and its autograd backward as ONE call each.  They issue exactly the launches of the per-op entries above (d2s_layernorm_fwd,
 * d2s_gemm_f32, d2s_attn_fwd_f32, ... in the same order with the same arguments: bit-identical results); what they remove is the host
 * cost of 7 + 11 separate calls per block, which bounds the step at the per-rank batches of the 8-GPU configurations (ddp_training.py:15:
 * 256 images over 8 ranks).  fp32 data path, arithmetic modes 0 / 1.
 * params: HOST array of 12 DEVICE pointers - norm1.weight, norm1.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias,
 * norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias (the state-dict order of a block). */
long d2s_block_saved_floats(int B, int n, int D, int H, int hidden, int train);   /* floats of the `saved` slab (train: what the backward needs) */
long d2s_block_bwd_scratch_floats(int B, int n, int D, int H, int hidden);
size_t d2s_block_workspace_bytes(int B, int n, int D, int hidden, int mode);       /* scratch of the launches on `stream` */
size_t d2s_block_wgrad_workspace_bytes(int B, int n, int D, int hidden, int mode); /* scratch of the weight-gradient launches */
/* x [B,n,D] -> y [B,n,D]; cls_row [B,H,n] or NULL = the CLS row of the softmax (:234); D = H * 64; train != 0 fills `saved` for the backward */
int d2s_block_fwd_f32(const float* x, const float* const* params, int B, int n, int D, int H, int hidden, float eps, float scale, float* y,
                      float* cls_row, float* saved, int train, int mode, void* workspace, size_t workspace_bytes, d2s_stream_t stream);
/* gy -> dx (NULL = not wanted) and dparams (HOST array of 12 device pointers in the order of params, NULL = not wanted; a LayerNorm's
 * weight / bias gradients come as a pair).  paramsT: NULL or HOST array {qkv.weight^T, proj.weight^T, fc1.weight^T, fc2.weight^T}
 * (d2s_transpose_batched_f32) for NT-layout input gradients, NULL entries = NN layout.  wgrad_stream: NULL or a second stream on which the
 * (dW, db) launches are issued after a fork from `stream` (scratch: wgrad_workspace); the caller joins the two streams before reading. */
int d2s_block_bwd_f32(const float* gy, const float* x, const float* saved, const float* const* params, const float* const* paramsT, int B,
                      int n, int D, int H, int hidden, float scale, float* dx, float* const* dparams, float* scratch, int mode, void* workspace,
                      size_t workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, d2s_stream_t wgrad_stream,
                      d2s_stream_t stream);

/* ---- optimiser: torch.optim.AdamW (mask_predictor.py:229-230) over a flat arena, one launch ------------------------ */
int d2s_adamw_chunk_elems(void);
/* chunk_steps (n_chunks ints, device memory, zero-initialised by the caller): torch.optim.AdamW's per-parameter state['step'] - a
 * tensor frozen during the warm-up epochs (utils.py:112-119) starts its bias correction at t = 1 when it is first updated; the call
 * advances the counters of the active chunks.  NULL: `step` (>= 1) is used for every chunk. */
int d2s_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const void* chunk_desc, int n_chunks,
                   float beta1, float beta2, float eps, int step, float grad_scale, int* chunk_steps, d2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* D2S_HIP_H */
