// Composite entry points: one transformer block (forward / backward) and one score predictor (forward / backward) per C-ABI call.
//
// Why: at the per-rank batches of the 8-GPU configurations (BASELINE config 3: 32 images per GPU) the step is bound by how fast the
// host can issue launches - ~450 C-ABI calls per step from Python autograd Functions, 12.5 ms of enqueue for ~10 ms of kernels
// (profiles/r03_a_graph_vs_eager.txt; a hipGraph replay of the same step is no faster on ROCm 7.2).  These entries issue exactly the
// launch sequence the per-op entries would - same kernels, same arguments, same order, hence bit-identical results - from C, so a block
// forward is 1 call instead of 7 (+ 7 tensor allocations and workspace queries on the Python side) and a block backward 1 instead of 11.
//
// Reference: vit_models/dynamic_vit.py:263-283 (Block.forward), :216-236 (Attention), :169-175 (Mlp) and their autograd backward;
// :536-551 with layers :491-531 (PredictorLG).  fp32 data path (arithmetic modes 0 and 1); the bf16 data path of mode 2 keeps its per-op
// entries (its operand forms differ per layer).
#include "d2s_common.h"

#include <mutex>

extern "C" {
size_t d2s_gemm_f32_workspace_bytes(int layout, int M, int N, int K, int mode);
int d2s_gemm_f32(int layout, const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K, int epilogue,
                 const float* bias, const float* aux, long ldaux, float* aux_out, int aux_rows, int remap_rows_per_img, int remap_skip,
                 int accumulate, int mode, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t d2s_linear_wgrad_workspace_bytes(int tokens, int n_out, int n_in, int mode);
int d2s_linear_wgrad_f32(const float* dy, long lddy, const float* x, long ldx, float* dW, long lddw, float* db, int tokens, int n_out,
                         int n_in, int accumulate, int mode, void* workspace, size_t workspace_bytes, hipStream_t stream);
int d2s_colsum_f32(const float* X, long ldx, int M, int N, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                   hipStream_t stream);
size_t d2s_colsum_workspace_bytes(int M, int N);
int d2s_layernorm_fwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w, const float* b,
                      float* y, float* mean, float* rstd, long rows, int D, float eps, hipStream_t stream);
size_t d2s_layernorm_bwd_workspace_bytes(long rows, int D);
int d2s_layernorm_bwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy, const float* w,
                      const float* mean, const float* rstd, float* dx, const float* add_src, float* dweight, float* dbias, int accumulate_wb,
                      int relu_mask, long rows, int D, void* workspace, size_t workspace_bytes, hipStream_t stream);
int d2s_attn_fwd_f32(const float* qkv, float* out, float* lse, float* cls_row, int B, int n, int H, float scale, hipStream_t stream);
int d2s_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta_ws, int B, int n,
                     int H, float scale, hipStream_t stream);
int d2s_half_mean_concat(const float* x, const float* relu_mask_src, float* out, int B, int T, int C, hipStream_t stream);
int d2s_softmax_rows(const float* scores, float* probs, int rows, int T, hipStream_t stream);
}

namespace {

enum { NT = 0, NN = 1 };
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_BIAS_GELU = 3, EPI_BIAS_RESID = 4, EPI_MUL_GELU_GRAD = 5 };
// parameter slots of a block (the order of Block._params() in vit_models/dynamic_vit.py)
enum { N1W, N1B, QKVW, QKVB, PROJW, PROJB, N2W, N2B, FC1W, FC1B, FC2W, FC2B, NPARAM };

inline long seg(long n) { return (n + 63) & ~63L; }      // every segment of a slab starts 256-byte aligned

// saved-for-backward slab of a block, in floats
struct BlockSaved {
    long ln1, mean1, rstd1, qkv, ao, lse, x1, ln2, mean2, rstd2, z, h, total;
    BlockSaved(long B, long n, long D, long H, long hid, bool train) {
        const long M = B * n;
        long o = 0;
        auto take = [&](long cnt) { const long at = o; o += seg(cnt); return at; };
        ln1 = take(M * D);
        qkv = take(M * 3 * D);
        ao = take(M * D);
        lse = take(B * H * n);
        x1 = take(M * D);
        h = take(M * hid);
        if (train) {
            ln2 = take(M * D);
            mean1 = take(M); rstd1 = take(M); mean2 = take(M); rstd2 = take(M);
            z = take(M * hid);
        } else {        // forward only: no statistics, no pre-activation copy, LayerNorm 2 reuses LayerNorm 1's buffer
            ln2 = ln1;
            mean1 = rstd1 = mean2 = rstd2 = z = -1;
        }
        total = o;
    }
};

// scratch slab of a block backward, in floats: dz, g1 and dqkv are read by the weight-gradient stream after the main stream has moved
// on, so they get buffers of their own; dln2 / dao / dln1 live and die on the main stream one after the other and share `t`
struct BlockBwdScratch {
    long dz, g1, dqkv, delta, t, total;
    BlockBwdScratch(long B, long n, long D, long H, long hid) {
        const long M = B * n;
        long o = 0;
        auto take = [&](long cnt) { const long at = o; o += seg(cnt); return at; };
        dz = take(M * hid);
        g1 = take(M * D);
        dqkv = take(M * 3 * D);
        delta = take(B * H * n);
        t = take(M * D);
        total = o;
    }
};

inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// ---- fork / join between the caller's stream and its weight-gradient stream ----
// The weight gradients of a Linear are read by nobody inside the backward pass, so they trail the dy -> dx chain on a second stream
// (DESIGN section 6).  A fork is "side waits for everything issued on main so far": one event record + one stream wait.  Events come from
// a small ring created on first use (host objects; no device memory is allocated by this library).
constexpr int NEV = 64;
hipEvent_t g_ev[NEV];
bool g_ev_ready = false;
std::mutex g_ev_mu;
unsigned g_ev_next = 0;

int fork_to(hipStream_t main, hipStream_t side) {
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        if (!g_ev_ready) {
            for (int i = 0; i < NEV; ++i)
                if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return D2S_ERR_LAUNCH;
            g_ev_ready = true;
        }
        ev = g_ev[g_ev_next++ % NEV];
    }
    if (hipEventRecord(ev, main) != hipSuccess) return D2S_ERR_LAUNCH;
    if (hipStreamWaitEvent(side, ev, 0) != hipSuccess) return D2S_ERR_LAUNCH;
    return D2S_OK;
}

#define D2S_TRY(expr)                 \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != D2S_OK) return rc_; \
    } while (0)

struct Ctx {
    int mode;
    void* ws; size_t ws_bytes;
    hipStream_t stream;
};

inline int linear_fwd(const Ctx& c, const float* x, const float* W, const float* b, float* y, int M, int N, int K, int epi,
                      const float* aux = nullptr, float* aux_out = nullptr) {
    return d2s_gemm_f32(NT, x, K, W, K, y, N, M, N, K, epi, b, aux, aux ? N : 0, aux_out, 0, 0, 0, 0, c.mode, c.ws, c.ws_bytes, c.stream);
}

// dx[M,K] = epi(dy[M,N] W[N,K]); with the k-contiguous copy W^T [K,N] at hand it runs as an NT product (exact mode, d2s.ops.linear_dgrad)
inline int linear_dgrad(const Ctx& c, const float* dy, const float* W, const float* Wt, float* dx, int M, int N, int K, int epi = EPI_NONE,
                        const float* aux = nullptr) {
    if (Wt) return d2s_gemm_f32(NT, dy, N, Wt, N, dx, K, M, K, N, epi, nullptr, aux, aux ? K : 0, nullptr, 0, 0, 0, 0, c.mode, c.ws, c.ws_bytes, c.stream);
    return d2s_gemm_f32(NN, dy, N, W, K, dx, K, M, K, N, epi, nullptr, aux, aux ? K : 0, nullptr, 0, 0, 0, 0, c.mode, c.ws, c.ws_bytes, c.stream);
}

struct Side {
    hipStream_t main, side;      // side == nullptr: weight gradients in line on the main stream
    void* ws; size_t ws_bytes;   // the side stream's own scratch (the main stream's when side == nullptr)
    int mode;
    bool used = false;
};

// (dW, db) of a Linear from dy [M, N] and its input x [M, K]; either may be unwanted
inline int param_grads(Side& s, const float* dy, const float* x, float* dW, float* db, int M, int N, int K) {
    if (!dW && !db) return D2S_OK;
    hipStream_t st = s.main;
    if (s.side) {
        D2S_TRY(fork_to(s.main, s.side));
        st = s.side;
        s.used = true;
    }
    if (dW) return d2s_linear_wgrad_f32(dy, N, x, K, dW, K, db, M, N, K, 0, s.mode, s.ws, s.ws_bytes, st);
    return d2s_colsum_f32(dy, N, M, N, db, 0, s.ws, s.ws_bytes, st);
}

}  // namespace

extern "C" {

// floats of the slab d2s_block_fwd_f32 fills: train != 0 -> everything the backward needs; 0 -> forward-only scratch
long d2s_block_saved_floats(int B, int n, int D, int H, int hidden, int train) { return BlockSaved(B, n, D, H, hidden, train != 0).total; }
long d2s_block_bwd_scratch_floats(int B, int n, int D, int H, int hidden) { return BlockBwdScratch(B, n, D, H, hidden).total; }

// scratch of the launches issued on the caller's stream (split-K slabs of small grids, LayerNorm-backward partials)
size_t d2s_block_workspace_bytes(int B, int n, int D, int hidden, int mode) {
    const int M = B * n;
    size_t w = 0;
    w = max2(w, d2s_gemm_f32_workspace_bytes(NT, M, 3 * D, D, mode));
    w = max2(w, d2s_gemm_f32_workspace_bytes(NT, M, D, D, mode));
    w = max2(w, d2s_gemm_f32_workspace_bytes(NT, M, hidden, D, mode));
    w = max2(w, d2s_gemm_f32_workspace_bytes(NT, M, D, hidden, mode));
    w = max2(w, d2s_gemm_f32_workspace_bytes(NT, M, D, 3 * D, mode));
    for (int lay = NT; lay <= NN; ++lay) {      // input gradients: NT through W^T, or NN
        w = max2(w, d2s_gemm_f32_workspace_bytes(lay, M, hidden, D, mode));
        w = max2(w, d2s_gemm_f32_workspace_bytes(lay, M, D, hidden, mode));
        w = max2(w, d2s_gemm_f32_workspace_bytes(lay, M, D, D, mode));
        w = max2(w, d2s_gemm_f32_workspace_bytes(lay, M, D, 3 * D, mode));
    }
    w = max2(w, d2s_layernorm_bwd_workspace_bytes(M, D));
    return w;
}
// scratch of the weight-gradient launches (the side stream's own buffer when there is one)
size_t d2s_block_wgrad_workspace_bytes(int B, int n, int D, int hidden, int mode) {
    const int M = B * n;
    size_t w = 0;
    w = max2(w, d2s_linear_wgrad_workspace_bytes(M, D, hidden, mode));
    w = max2(w, d2s_linear_wgrad_workspace_bytes(M, hidden, D, mode));
    w = max2(w, d2s_linear_wgrad_workspace_bytes(M, D, D, mode));
    w = max2(w, d2s_linear_wgrad_workspace_bytes(M, 3 * D, D, mode));
    w = max2(w, d2s_colsum_workspace_bytes(M, hidden));
    w = max2(w, d2s_colsum_workspace_bytes(M, 3 * D));
    return w;
}

// y = Block(x): x + proj(attn(LN1 x)), then + fc2(gelu(fc1(LN2 .)))  on a packed [B, n, D] token tensor.
//   params: HOST array of 12 device pointers (norm1.weight, norm1.bias, qkv.weight, qkv.bias, proj.weight, proj.bias, norm2.weight,
//           norm2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
//   cls_row [B, H, n] or NULL: the CLS row of the softmax (Attention.forward's second output, :234)
//   saved: slab of d2s_block_saved_floats(.., train) floats; train != 0: it holds what d2s_block_bwd_f32 needs afterwards
int d2s_block_fwd_f32(const float* x, const float* const* params, int B, int n, int D, int H, int hidden, float eps, float scale, float* y,
                      float* cls_row, float* saved, int train, int mode, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !params || !y || !saved || B <= 0 || n <= 0 || D <= 0 || H <= 0 || hidden <= 0 || D != H * 64 || mode < 0 || mode > 1) return D2S_ERR_ARG;
    for (int i = 0; i < NPARAM; ++i)      // the four Linear biases are optional (T2T blocks: qkv_bias=False, transformer_block.py:31)
        if (!params[i] && i != QKVB && i != PROJB && i != FC1B && i != FC2B) return D2S_ERR_ARG;
    const int M = B * n;
    const BlockSaved L(B, n, D, H, hidden, train != 0);
    const Ctx c{mode, workspace, workspace_bytes, stream};
    float* ln1 = saved + L.ln1;
    float* qkv = saved + L.qkv;
    float* ao = saved + L.ao;
    float* x1 = saved + L.x1;
    float* ln2 = saved + L.ln2;
    float* h = saved + L.h;
    D2S_TRY(d2s_layernorm_fwd(x, M, 0, D, 0, params[N1W], params[N1B], ln1, train ? saved + L.mean1 : nullptr, train ? saved + L.rstd1 : nullptr,
                              M, D, eps, stream));
    D2S_TRY(linear_fwd(c, ln1, params[QKVW], params[QKVB], qkv, M, 3 * D, D, params[QKVB] ? EPI_BIAS : EPI_NONE));
    D2S_TRY(d2s_attn_fwd_f32(qkv, ao, saved + L.lse, cls_row, B, n, H, scale, stream));
    D2S_TRY(linear_fwd(c, ao, params[PROJW], params[PROJB], x1, M, D, D, EPI_BIAS_RESID, x));
    D2S_TRY(d2s_layernorm_fwd(x1, M, 0, D, 0, params[N2W], params[N2B], ln2, train ? saved + L.mean2 : nullptr, train ? saved + L.rstd2 : nullptr,
                              M, D, eps, stream));
    D2S_TRY(linear_fwd(c, ln2, params[FC1W], params[FC1B], h, M, hidden, D, EPI_BIAS_GELU, nullptr, train ? saved + L.z : nullptr));
    D2S_TRY(linear_fwd(c, h, params[FC2W], params[FC2B], y, M, D, hidden, EPI_BIAS_RESID, x1));
    return D2S_OK;
}

// Backward of d2s_block_fwd_f32 (train slab).  gy [B, n, D] -> dx [B, n, D] (NULL: not wanted, the attention / LayerNorm-1 leg is then
// skipped unless a LayerNorm-1 gradient is wanted, in which case dx must be given) and the parameter gradients:
//   dparams: HOST array of 12 device pointers in the order of `params`, NULL = not wanted (a LayerNorm's weight and bias come as a pair)
//   paramsT: HOST array of 4 device pointers {qkv.weight^T, proj.weight^T, fc1.weight^T, fc2.weight^T} (k-contiguous copies for the
//            input-gradient GEMMs, d2s_transpose_batched_f32) - NULL array or NULL entries: the NN layout on the weight itself
//   scratch: d2s_block_bwd_scratch_floats floats
//   wgrad_stream: NULL = weight gradients in line; else every Linear's (dW, db) launch is issued there, after a fork from `stream`, with
//            wgrad_workspace as scratch - the caller joins the two streams before anything reads a gradient
int d2s_block_bwd_f32(const float* gy, const float* x, const float* saved, const float* const* params, const float* const* paramsT, int B,
                      int n, int D, int H, int hidden, float scale, float* dx, float* const* dparams, float* scratch, int mode, void* workspace,
                      size_t workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, hipStream_t wgrad_stream,
                      hipStream_t stream) {
    if (!gy || !x || !saved || !params || !dparams || !scratch || B <= 0 || n <= 0 || D <= 0 || H <= 0 || hidden <= 0 || D != H * 64 || mode < 0 || mode > 1)
        return D2S_ERR_ARG;
    if ((dparams[N1W] == nullptr) != (dparams[N1B] == nullptr) || (dparams[N2W] == nullptr) != (dparams[N2B] == nullptr)) return D2S_ERR_ARG;
    const bool leg1 = dx || dparams[N1W];
    if (leg1 && !dx) return D2S_ERR_ARG;
    const int M = B * n;
    const BlockSaved L(B, n, D, H, hidden, true);
    const BlockBwdScratch S(B, n, D, H, hidden);
    const Ctx c{mode, workspace, workspace_bytes, stream};
    Side side{stream, wgrad_stream, wgrad_stream ? wgrad_workspace : workspace, wgrad_stream ? wgrad_workspace_bytes : workspace_bytes, mode};
    const float* qkvT = paramsT ? paramsT[0] : nullptr;
    const float* projT = paramsT ? paramsT[1] : nullptr;
    const float* fc1T = paramsT ? paramsT[2] : nullptr;
    const float* fc2T = paramsT ? paramsT[3] : nullptr;
    float* dz = scratch + S.dz;
    float* g1 = scratch + S.g1;
    float* dqkv = scratch + S.dqkv;
    float* t = scratch + S.t;
    // ---- MLP branch ----
    D2S_TRY(param_grads(side, gy, saved + L.h, dparams[FC2W], dparams[FC2B], M, D, hidden));
    D2S_TRY(linear_dgrad(c, gy, params[FC2W], fc2T, dz, M, D, hidden, EPI_MUL_GELU_GRAD, saved + L.z));
    D2S_TRY(param_grads(side, dz, saved + L.ln2, dparams[FC1W], dparams[FC1B], M, hidden, D));
    D2S_TRY(linear_dgrad(c, dz, params[FC1W], fc1T, t, M, hidden, D));                                   // t = dln2
    D2S_TRY(d2s_layernorm_bwd(saved + L.x1, M, 0, D, 0, t, params[N2W], saved + L.mean2, saved + L.rstd2, g1, gy, dparams[N2W], dparams[N2B], 0, 0,
                              M, D, workspace, workspace_bytes, stream));
    // ---- attention branch ----
    D2S_TRY(param_grads(side, g1, saved + L.ao, dparams[PROJW], dparams[PROJB], M, D, D));
    D2S_TRY(linear_dgrad(c, g1, params[PROJW], projT, t, M, D, D));                                      // t = dao
    D2S_TRY(d2s_attn_bwd_f32(saved + L.qkv, saved + L.ao, t, saved + L.lse, dqkv, scratch + S.delta, B, n, H, scale, stream));
    D2S_TRY(param_grads(side, dqkv, saved + L.ln1, dparams[QKVW], dparams[QKVB], M, 3 * D, D));
    if (leg1) {
        D2S_TRY(linear_dgrad(c, dqkv, params[QKVW], qkvT, t, M, 3 * D, D));                              // t = dln1
        D2S_TRY(d2s_layernorm_bwd(x, M, 0, D, 0, t, params[N1W], saved + L.mean1, saved + L.rstd1, dx, g1, dparams[N1W], dparams[N1B], 0, 0, M, D,
                                  workspace, workspace_bytes, stream));
    }
    return D2S_OK;
}

}  // extern "C"
