"""Autograd glue: one torch.autograd.Function per module of the hot path.  Forward and backward are explicit
sequences of C-ABI calls (d2s.ops); torch only owns the tensors and the graph.  No torch arithmetic runs here.

Reference lines (relative to /root/reference):
  EmbedFn      vit_models/dynamic_vit.py:300-306, 820-823
  BlockFn      vit_models/dynamic_vit.py:263-269 (Block), :216-236 (Attention), :169-175 (Mlp)
  PredictorFn  vit_models/dynamic_vit.py:536-551 with layers :491-531
  GatherFn     vit_models/dynamic_vit.py:907-912
  HeadFn       vit_models/dynamic_vit.py:993-1006
"""
import threading
import weakref

import torch

from . import ops

_DH = 64


def _zeros_like_param(p):
    return torch.zeros_like(p)


def _need(ctx, i):
    return ctx.needs_input_grad[i]


_tls = threading.local()


def run(fn, *args):
    """fn.apply(*args) with the grad mode decided HERE: inside Function.forward grad mode is always off and ctx.needs_input_grad only
    reflects the inputs' requires_grad, so under torch.no_grad() a trainable model (the student in evaluate.py) would take the
    training path - LayerNorm statistics, the GELU pre-activation copy, save_for_backward.  The flag makes the forward-only fast paths
    of BlockFn / EmbedFn / PredictorFn / HeadFn the ones that run (the tensors themselves are passed unchanged: detached aliases would
    defeat the identity checks of the weight caches in d2s.ops)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    prev = getattr(_tls, "no_grad", False)
    _tls.no_grad = True
    try:
        return fn.apply(*args)
    finally:
        _tls.no_grad = prev


def wants_grad(ctx):
    """Does this forward have to keep anything for a backward?"""
    return any(ctx.needs_input_grad) and not getattr(_tls, "no_grad", False)


def mode_recorded(cls):
    """Class decorator for Functions that launch GEMM / attention kernels: the forward records the arithmetic mode it ran in
    (ops.get_gemm_mode(): the process default or the caller's `with ops.gemm_mode(...)`), the backward - which autograd runs on its
    own thread, outside the caller's block - re-enters it, so forward and backward of one module always use the same arithmetic."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx.gmode = ops.get_gemm_mode()
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        with ops.gemm_mode(ctx.gmode):
            return bwd(ctx, *grads)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


# Teacher and student embed the SAME image batch (train.py:40,43): inside `shared_patch_columns()` the im2col matrix of an image
# tensor is built once and handed to every EmbedFn that asks for it (same tensor object, same version counter, same patch size,
# same stream).  The block is what bounds the cache's lifetime - nothing is kept once it exits.
_COLS = {"depth": 0, "entry": None}


class shared_patch_columns:
    def __enter__(self):
        _COLS["depth"] += 1
        return self

    def __exit__(self, *exc):
        _COLS["depth"] -= 1
        if _COLS["depth"] == 0:
            _COLS["entry"] = None
        return False


def _patch_columns(img, patch):
    if _COLS["depth"] == 0:
        return ops.im2col_patch(img, patch)
    stream = torch.cuda.current_stream(img.device).cuda_stream
    ent = _COLS["entry"]
    if ent is not None and ent[0]() is img and ent[1:3] == (img._version, patch) and ent[3] in (stream, None):
        return ent[4]
    col = ops.im2col_patch(img, patch)
    _COLS["entry"] = (weakref.ref(img), img._version, patch, stream, col)
    return col


def prime_patch_columns(img, patch):
    """Build the im2col matrix of `img` now, on the current stream, and mark it usable from ANY stream: for a caller that then forks
    work onto a second stream which first waits on this one (d2s.engine.TrainStep with the teacher on its own stream)."""
    if _COLS["depth"] == 0:
        return
    img = img.contiguous()
    _COLS["entry"] = (weakref.ref(img), img._version, patch, None, ops.im2col_patch(img, patch))


@mode_recorded
class EmbedFn(torch.autograd.Function):
    """img [B,3,H,W] -> tokens [B, T+1, D]  (conv-as-GEMM with the bias + pos_embed add fused in the epilogue)."""

    @staticmethod
    def forward(ctx, img, proj_w, proj_b, cls_token, pos_embed, patch):
        B = img.shape[0]
        D = proj_w.shape[0]
        col = _patch_columns(img.contiguous(), patch)
        T = col.shape[0] // B
        Kc = col.shape[1]
        tokens = torch.empty((B, T + 1, D), dtype=torch.float32, device=img.device)
        w2 = proj_w.reshape(D, Kc)
        pos = pos_embed.reshape(T + 1, D)
        ops.gemm(ops.NT, col, Kc, w2, Kc, tokens, D, B * T, D, Kc, ops.EPI_BIAS_ROWADD, proj_b, pos[1:], D, None, T, T, 1)
        ops.fill_cls(cls_token.reshape(D), pos, tokens)
        if wants_grad(ctx):       # forward-only callers (frozen teacher, eval) keep nothing
            ctx.save_for_backward(col, proj_w, proj_b, cls_token, pos_embed)
            ctx.dims = (B, T, D, Kc, tuple(proj_w.shape))
        return tokens

    @staticmethod
    def backward(ctx, g):
        col, proj_w, proj_b, cls_token, pos_embed = ctx.saved_tensors
        B, T, D, Kc, wshape = ctx.dims
        g = g.contiguous()
        dw = db = dcls = dpos = None
        if _need(ctx, 1) or _need(ctx, 2):
            gp = ops.copy_rows(g, ops.skip_cls_map(T + 1, D), B * T, D)
            db = ops.grad_buffer(proj_b) if _need(ctx, 2) else None
            if _need(ctx, 1):
                dw = ops.grad_buffer(proj_w)
                ops.linear_wgrad(gp, col, dw.view(D, Kc), db=db)
            elif db is not None:
                ops.colsum(gp, db)
        if _need(ctx, 3) or _need(ctx, 4):
            dpos = ops.grad_buffer(pos_embed)
            ops.batch_sum(g, dpos, B, (T + 1) * D, (T + 1) * D)
            if _need(ctx, 3):
                dcls = ops.batch_sum(g, ops.grad_buffer(cls_token), B, D, (T + 1) * D)
            if not _need(ctx, 4):
                dpos = None
        return None, dw, db, dcls, dpos, None


@mode_recorded
class BlockFn(torch.autograd.Function):
    """One pre-norm transformer block on a packed [B, n, D] token tensor; also returns the CLS row of the softmax."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, heads, eps, want_cls, scale, *extra):
        policy = extra[0] if extra else None         # optional 18th input: keep policy [B, n] of the dynamic-keep-ratio path
        ctx.nextra = len(extra)
        B, n, D = x.shape
        M = B * n
        x = x.contiguous()
        scale = float(_DH) ** -0.5 if scale is None else float(scale)     # Attention.scale = qk_scale or head_dim ** -0.5 (:188)
        cmap = ops.contiguous_map(M, D)
        hidden = fc1w.shape[0]
        # bf16 arithmetic mode: LayerNorm, the attention forward and the fc1 epilogue also emit the bf16 form of what the next GEMM
        # multiplies, and that GEMM reads it instead of converting its fp32 operand (no conversion passes on the forward path)
        io = ops.bf16_io() and D % 32 == 0 and hidden % 32 == 0 and x.is_cuda
        io_attn = io and policy is None and ops._BF16_ATTENTION
        ops._SHADOW.clear()          # gradient shadows never outlive the backward pass that made them
        ctx.composite = False
        if policy is None and not io and ops.block_composite_ok(x, heads, hidden):
            # fp32 data path: the whole block is ONE C-ABI call (csrc/block.hip issues the same seven launches)
            train = wants_grad(ctx)
            params = (n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b)
            y, cls_row, slab = ops.block_fwd(x, params, B, n, D, heads, hidden, eps, scale, want_cls, train)
            if train:
                ctx.save_for_backward(x, slab, *params)
                ctx.composite = True
                ctx.dims = (B, n, D, heads, scale)
                ctx.hidden = hidden
            if cls_row is None:
                cls_row = torch.empty((0,), device=x.device)
            ctx.mark_non_differentiable(cls_row)
            return y, cls_row
        if not wants_grad(ctx):
            # forward-only (the frozen teacher under no_grad, eval): no LayerNorm statistics, no GELU pre-activation copy (155 MB per
            # block at B=128), nothing saved; on the bf16 data path not even the fp32 form of the GEMM inputs
            if io:
                _, _, _, ln1h = ops.layernorm_fwd_bf16(x, cmap, n1w, n1b, M, D, eps, stats=False, want_f32=False)
                if io_attn:      # the bf16 attention kernel rounds q, k, v to bf16 anyway: the qkv GEMM writes only that form
                    qkv = ops.bf16_buffer(M, 3 * D, x.device)
                    ops.linear_fwd(None, qkvw, qkvb, a16=ln1h, c16=qkv, want_f32=False)
                else:
                    qkv = ops.linear_fwd(None, qkvw, qkvb, a16=ln1h)
                del ln1h
                if io_attn:
                    ao, _, cls_row, aoh = ops.attn_fwd_bf16io(qkv, B, n, heads, scale, want_cls, want_f32=False)
                elif policy is None:
                    (ao, _, cls_row), aoh = ops.attn_fwd(qkv, B, n, heads, scale, want_cls), None
                else:
                    (ao, _, _, cls_row), aoh = ops.attn_policy_fwd(qkv, policy, B, n, heads, scale, want_cls=want_cls), None
                del qkv
                x1 = ops.linear_fwd(ao, projw, projb, epi=ops.EPI_BIAS_RESID, aux=x.view(M, D), a16=aoh)
                del ao, aoh
                _, _, _, ln2h = ops.layernorm_fwd_bf16(x1, cmap, n2w, n2b, M, D, eps, stats=False, want_f32=False)
                hh = ops.bf16_buffer(M, hidden, x.device)
                ops.linear_fwd(None, fc1w, fc1b, epi=ops.EPI_BIAS_GELU, a16=ln2h, c16=hh, want_f32=False)
                del ln2h
                y = ops.linear_fwd(None, fc2w, fc2b, epi=ops.EPI_BIAS_RESID, aux=x1, a16=hh)
                if cls_row is None:
                    cls_row = torch.empty((0,), device=x.device)
                return y.view(B, n, D), cls_row
            ln1, _, _ = ops.layernorm_fwd(x, cmap, n1w, n1b, M, D, eps, stats=False)
            qkv = ops.linear_fwd(ln1, qkvw, qkvb)
            del ln1
            if policy is None:
                ao, _, cls_row = ops.attn_fwd(qkv, B, n, heads, scale, want_cls)
            else:
                ao, _, _, cls_row = ops.attn_policy_fwd(qkv, policy, B, n, heads, scale, want_cls=want_cls)
            del qkv
            x1 = ops.linear_fwd(ao, projw, projb, epi=ops.EPI_BIAS_RESID, aux=x.view(M, D))
            ln2, _, _ = ops.layernorm_fwd(x1, cmap, n2w, n2b, M, D, eps, stats=False)
            h = ops.linear_fwd(ln2, fc1w, fc1b, epi=ops.EPI_BIAS_GELU)
            y = ops.linear_fwd(h, fc2w, fc2b, epi=ops.EPI_BIAS_RESID, aux=x1)
            if cls_row is None:
                cls_row = torch.empty((0,), device=x.device)
            return y.view(B, n, D), cls_row
        ln1h = aoh = ln2h = hh = None
        if io:      # the GEMMs read the bf16 forms and so do the weight gradients: LayerNorm outputs and the GELU activation are kept in bf16 only
            _, mean1, rstd1, ln1h = ops.layernorm_fwd_bf16(x, cmap, n1w, n1b, M, D, eps, want_f32=False)
            ln1 = ln1h
        else:
            ln1, mean1, rstd1 = ops.layernorm_fwd(x, cmap, n1w, n1b, M, D, eps)
        if io_attn:          # bf16 qkv only (see the forward-only branch); saved for the backward in that form
            qkv = ops.bf16_buffer(M, 3 * D, x.device)
            ops.linear_fwd(None, qkvw, qkvb, a16=ln1h, c16=qkv, want_f32=False)
        elif io:
            qkv = ops.linear_fwd(None, qkvw, qkvb, a16=ln1h)
        else:
            qkv = ops.linear_fwd(ln1, qkvw, qkvb)
        del ln1h
        cinv = None
        if io_attn:
            ao, lse, cls_row, aoh = ops.attn_fwd_bf16io(qkv, B, n, heads, scale, want_cls)
        elif policy is None:
            ao, lse, cls_row = ops.attn_fwd(qkv, B, n, heads, scale, want_cls)
        else:   # dynamic keep ratio: softmax_with_policy fused into the attention pass (:195-214)
            ao, lse, cinv, cls_row = ops.attn_policy_fwd(qkv, policy, B, n, heads, scale, want_cls=want_cls)
        x2d = x.view(M, D)
        x1 = ops.linear_fwd(ao, projw, projb, epi=ops.EPI_BIAS_RESID, aux=x2d, a16=aoh)
        # GELU pre-activation for the backward: fp32, or bf16 on the bf16 data path (what autocast keeps: fc1's output is bf16 there)
        z = torch.empty((M, hidden), dtype=torch.bfloat16 if (io and ops._BF16_PREACT) else torch.float32, device=x.device)
        if io:
            _, mean2, rstd2, ln2h = ops.layernorm_fwd_bf16(x1, cmap, n2w, n2b, M, D, eps, want_f32=False)
            ln2 = ln2h
            hh = ops.bf16_buffer(M, hidden, x.device)
            ops.linear_fwd(None, fc1w, fc1b, epi=ops.EPI_BIAS_GELU, aux_out=z, a16=ln2h, c16=hh, want_f32=False)
            h = hh
            y = ops.linear_fwd(None, fc2w, fc2b, epi=ops.EPI_BIAS_RESID, aux=x1, a16=hh)
        else:
            ln2, mean2, rstd2 = ops.layernorm_fwd(x1, cmap, n2w, n2b, M, D, eps)
            h = ops.linear_fwd(ln2, fc1w, fc1b, epi=ops.EPI_BIAS_GELU, aux_out=z)
            y = ops.linear_fwd(h, fc2w, fc2b, epi=ops.EPI_BIAS_RESID, aux=x1)
        del ln2h, hh
        ctx.save_for_backward(x, n1w, qkvw, projw, n2w, fc1w, fc2w, mean1, rstd1, ln1, qkv, ao, lse, x1, mean2, rstd2, ln2, z, h,
                              n1b, qkvb, projb, n2b, fc1b, fc2b, aoh)      # aoh: bf16 form of ao (bf16 attention only), proj's weight gradient
        ctx.policy = (policy, cinv)
        ctx.dims = (B, n, D, heads, scale)
        if cls_row is None:
            cls_row = torch.empty((0,), device=x.device)
        ctx.mark_non_differentiable(cls_row)
        return y.view(B, n, D), cls_row

    @staticmethod
    def _backward_composite(ctx, gy):
        x, slab = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        B, n, D, heads, scale = ctx.dims
        wants = [_need(ctx, i) for i in range(13)]
        dparams = [None] * 12
        for i in range(12):
            pair = i ^ 1 if i in (0, 1, 6, 7) else i          # a LayerNorm's weight and bias gradients are produced together
            if wants[1 + i] or wants[1 + pair]:
                dparams[i] = ops.grad_buffer(params[i])
        want_dx = wants[0] or dparams[0] is not None
        dx = ops.block_bwd(gy.contiguous(), x, slab, params, B, n, D, heads, ctx.hidden, scale, want_dx, dparams)
        grads = [dx if wants[0] else None] + [dparams[i] if wants[1 + i] else None for i in range(12)]
        return tuple(grads) + (None, None, None, None) + (None,) * ctx.nextra

    @staticmethod
    def backward(ctx, gy, _gcls):
        if ctx.composite:
            return BlockFn._backward_composite(ctx, gy)
        (x, n1w, qkvw, projw, n2w, fc1w, fc2w, mean1, rstd1, ln1, qkv, ao, lse, x1, mean2, rstd2, ln2, z, h,
         n1b, qkvb, projb, n2b, fc1b, fc2b, aoh) = ctx.saved_tensors
        B, n, D, heads, scale = ctx.dims
        M = B * n
        dev = gy.device
        gy = gy.contiguous().view(M, D)
        cmap = ops.contiguous_map(M, D)
        # parameter order: n1w n1b qkvw qkvb projw projb n2w n2b fc1w fc1b fc2w fc2b -> input slots 1..12
        wants = [_need(ctx, i) for i in range(13)]
        grads = [None] * 13

        new = ops.grad_buffer

        # ---- MLP branch ----
        def xarg(t):      # a saved layer input is fp32, or bf16 only on the bf16 data path
            return (None, t) if t.dtype == torch.bfloat16 else (t, None)
        hx, h16 = xarg(h)
        # bf16 data path: every gradient that feeds an input-gradient GEMM is also produced in bf16 by the kernel that computes it
        io = ops.bf16_io() and z.shape[1] % 32 == 0 and D % 32 == 0 and gy.is_cuda
        policy, cinv = ctx.policy
        io_attn = io and policy is None and ops._BF16_ATTENTION
        gyh = ops.shadow_take(gy) if io else None
        if gyh is not None:
            gyh = gyh.view(M, D)
        # with both operands at hand in bf16 the weight-gradient kernel reads them where they lie (token-major, no transposing pass); the
        # bias gradient is then the sum of the bf16 gradient values, as under torch.autocast
        grads[11], grads[12] = ops.linear_param_grads(gy, hx, fc2w, fc2b, wants[11], wants[12], x16=h16, dy16=gyh if h16 is not None else None)
        dzh = ops.bf16_buffer(M, z.shape[1], dev) if io else None
        l2x, l216 = xarg(ln2)
        # dz = (gy fc2w) * gelu'(z) feeds two bf16 GEMMs and nothing else: on the bf16 data path only its bf16 form is written (fc1's bias
        # gradient is then the fp32 sum of those bf16 values, as under torch.autocast)
        dz_bf16_only = dzh is not None and l216 is not None and wants[9]
        dz = ops.linear_dgrad(gy, fc2w, epi=ops.EPI_MUL_GELU_GRAD, aux=z, a16=gyh, c16=dzh, want_f32=not dz_bf16_only)
        del gyh
        grads[9], grads[10] = ops.linear_param_grads(dz, l2x, fc1w, fc1b, wants[9], wants[10], x16=l216, dy16=dzh if dz_bf16_only else None)
        dln2 = ops.linear_dgrad(dz, fc1w, a16=dzh)
        del dzh
        g1 = torch.empty((M, D), dtype=torch.float32, device=dev)
        g1h = ops.bf16_buffer(M, D, dev) if io else None
        dn2w = new(n2w) if (wants[7] or wants[8]) else None
        dn2b = new(n2b) if dn2w is not None else None
        ops.layernorm_bwd(x1, cmap, dln2, n2w, mean2, rstd2, g1, gy, dn2w, dn2b, M, D, dx16=g1h)
        grads[7], grads[8] = (dn2w if wants[7] else None), (dn2b if wants[8] else None)
        # ---- attention branch ----
        grads[5], grads[6] = ops.linear_param_grads(g1, ao, projw, projb, wants[5], wants[6], x16=aoh, dy16=g1h if aoh is not None else None)
        dao = ops.linear_dgrad(g1, projw, a16=g1h)
        del g1h
        dqkvh = None
        l1x, l116 = xarg(ln1)
        dqkv_bf16_only = False
        if policy is None:
            if io_attn:
                dqkvh = torch.empty(qkv.shape, dtype=torch.bfloat16, device=dev)
                dqkv_bf16_only = l116 is not None and wants[3]      # both consumers (weight gradient, input gradient) read the bf16 form
            dqkv = ops.attn_bwd(qkv, ao, dao, lse, B, n, heads, scale, dqkv16=dqkvh, want_f32=not dqkv_bf16_only)
        else:
            dqkv = ops.attn_policy_bwd(qkv, policy, ao, dao, lse, cinv, B, n, heads, scale)
        grads[3], grads[4] = ops.linear_param_grads(dqkv, l1x, qkvw, qkvb, wants[3], wants[4], x16=l116, dy16=dqkvh if dqkv_bf16_only else None)
        gx = None
        if wants[0] or wants[1] or wants[2]:
            dln1 = ops.linear_dgrad(dqkv, qkvw, a16=dqkvh)
            del dqkvh
            gx = torch.empty((M, D), dtype=torch.float32, device=dev)
            gxh = ops.bf16_buffer(M, D, dev) if (io and wants[0]) else None
            dn1w = new(n1w) if (wants[1] or wants[2]) else None
            dn1b = new(n1b) if dn1w is not None else None
            ops.layernorm_bwd(x, cmap, dln1, n1w, mean1, rstd1, gx, g1, dn1w, dn1b, M, D, dx16=gxh)
            grads[1], grads[2] = (dn1w if wants[1] else None), (dn1b if wants[2] else None)
            gx = gx.view(B, n, D) if wants[0] else None
            if gxh is not None:
                ops.shadow_put(gx, gxh)
        grads[0] = gx
        return tuple(grads) + (None, None, None, None) + (None,) * ctx.nextra


@mode_recorded
class PredictorFn(torch.autograd.Function):
    """Large LayerNorm predictor (PredictorLG, topk_selection=True) on x[:, 1:] of a [B, n, D] tensor.
    Returns (scores [B, n-1] differentiable, keep_probs [B, n-1] non-differentiable)."""

    @staticmethod
    def forward(ctx, x, *params):
        # params: in_ln_w, in_ln_b, in_fc_w, in_fc_b, then 5 x (ln_w, ln_b, fc_w, fc_b)
        B, n, D = x.shape
        T = n - 1
        M = B * T
        x = x.contiguous()
        eps = 1e-5
        train = wants_grad(ctx)
        # bf16 arithmetic mode: as in BlockFn, each LayerNorm in front of a bf16-mode Linear also (training: only) writes the bf16 form
        # that GEMM multiplies; the weight gradients read it too.  The exact-fp32 tail keeps fp32 activations.
        io = ops.bf16_io() and x.is_cuda and D % 32 == 0
        if io:
            h0, mean0, rstd0, h0h = ops.layernorm_fwd_bf16(x, ops.skip_cls_map(n, D), params[0], params[1], M, D, eps, stats=train, want_f32=False)
            a1 = ops.linear_fwd(None, params[2], params[3], epi=ops.EPI_BIAS_RELU, a16=h0h)
            h0 = h0h
        else:
            h0, mean0, rstd0 = ops.layernorm_fwd(x, ops.skip_cls_map(n, D), params[0], params[1], M, D, eps, stats=train)
            a1 = ops.linear_fwd(h0, params[2], params[3], epi=ops.EPI_BIAS_RELU)
        C = a1.shape[1]
        cur = ops.half_mean_concat(a1, B, T, C)
        saved = [x, h0, mean0, rstd0, a1]
        nl = (len(params) - 4) // 4
        for j in range(nl):
            lw, lb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            width = cur.shape[1]
            last = j == nl - 1
            exact_tail = j >= nl - 2
            # the predictor's tail (its last two Linear layers, D/2 -> D/4 -> 1: 0.3 % of its FLOPs), like the softmax and the selection
            # behind it, always runs in exact fp32 - also in the bf16 arithmetic mode (kept-id stability)
            if io and not exact_tail and width % 32 == 0:
                _, mean, rstd, ln = ops.layernorm_fwd_bf16(cur, ops.contiguous_map(M, width), lw, lb, M, width, eps, stats=train, want_f32=False)
                nxt = ops.linear_fwd(None, fw, fb, epi=ops.EPI_BIAS_RELU, a16=ln)
            else:
                ln, mean, rstd = ops.layernorm_fwd(cur, ops.contiguous_map(M, width), lw, lb, M, width, eps, stats=train)
                with ops.gemm_mode(ops.GEMM_EXACT if exact_tail else ops.get_gemm_mode()):
                    nxt = ops.linear_fwd(ln, fw, fb, epi=ops.EPI_BIAS if last else ops.EPI_BIAS_RELU)
            saved += [cur, ln, mean, rstd]
            cur = nxt
        scores = cur.view(B, T)
        probs = ops.softmax_rows(scores)
        if train:
            ctx.save_for_backward(*saved, *params)
            ctx.meta = (B, n, D, T, M, C, nl, len(saved))
        ctx.mark_non_differentiable(probs)
        return scores, probs

    @staticmethod
    def backward(ctx, gscores, _gprobs):
        B, n, D, T, M, C, nl, nsaved = ctx.meta
        saved = ctx.saved_tensors[:nsaved]
        params = ctx.saved_tensors[nsaved:]
        x, h0, mean0, rstd0, a1 = saved[:5]
        dev = gscores.device
        np_ = len(params)
        grads = [None] * np_
        want = [_need(ctx, 1 + i) for i in range(np_)]
        d = gscores.contiguous().view(M, 1)
        d16 = None                      # bf16 copy of d, written by the LayerNorm backward that produced it, when the next GEMMs run in bf16
        io = ops.bf16_io() and gscores.is_cuda

        def xarg(t):                    # a saved layer input: fp32, or bf16 only (bf16 data path)
            return (None, t) if t.dtype == torch.bfloat16 else (t, None)
        for j in reversed(range(nl)):
            cur, ln, mean, rstd = saved[5 + 4 * j: 9 + 4 * j]
            lw, lb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            base = 4 + 4 * j
            width = cur.shape[1]
            lnx, ln16 = xarg(ln)
            # d is the gradient w.r.t. the pre-activation of layer j's Linear (the ReLU mask was applied upstream)
            with ops.gemm_mode(ops.GEMM_EXACT if j >= nl - 2 else ops.get_gemm_mode()):      # same arithmetic as the forward of this layer
                grads[base + 2], grads[base + 3] = ops.linear_param_grads(d, lnx, fw, fb, want[base + 2], want[base + 3], x16=ln16)
                dln = ops.linear_dgrad(d, fw, a16=d16 if ln16 is not None else None)
            dcur = torch.empty((M, width), dtype=torch.float32, device=dev)
            # the layer below (j - 1) multiplies dcur in bf16 if it is a bf16-mode layer: its saved input is bf16 then
            below_bf16 = io and j >= 1 and saved[5 + 4 * (j - 1) + 1].dtype == torch.bfloat16 and width % 32 == 0
            d16 = ops.bf16_buffer(M, width, dev) if below_bf16 else None
            dlw = ops.grad_buffer(lw) if (want[base] or want[base + 1]) else None
            dlb = ops.grad_buffer(lb) if dlw is not None else None
            # cur is the ReLU output of layer j-1 for j >= 1 -> fold that ReLU's backward in; for j == 0 cur is the
            # split/mean/concat output and the mask is applied by half_mean_concat below
            ops.layernorm_bwd(cur, ops.contiguous_map(M, width), dln, lw, mean, rstd, dcur, None, dlw, dlb, M, width,
                              relu_mask=(j >= 1), dx16=d16)
            grads[base], grads[base + 1] = (dlw if want[base] else None), (dlb if want[base + 1] else None)
            d = dcur
        dz1 = ops.half_mean_concat(d, B, T, C, relu_mask_src=a1)
        h0x, h016 = xarg(h0)
        grads[2], grads[3] = ops.linear_param_grads(dz1, h0x, params[2], params[3], want[2], want[3], x16=h016)
        gx = None
        if _need(ctx, 0) or want[0] or want[1]:
            dh0 = ops.linear_dgrad(dz1, params[2])
            gx = torch.zeros((B, n, D), dtype=torch.float32, device=dev)
            dlw = ops.grad_buffer(params[0]) if (want[0] or want[1]) else None
            dlb = ops.grad_buffer(params[1]) if dlw is not None else None
            ops.layernorm_bwd(x, ops.skip_cls_map(n, D), dh0, params[0], mean0, rstd0, gx, None, dlw, dlb, M, D)
            grads[0], grads[1] = (dlw if want[0] else None), (dlb if want[1] else None)
            if not _need(ctx, 0):
                gx = None
        return (gx,) + tuple(grads)


class GatherFn(torch.autograd.Function):
    """Pack the surviving tokens: out[b] = x[b, [0, kept+1]]; backward scatters (zeros for dropped tokens)."""

    @staticmethod
    def forward(ctx, x, kept):
        ctx.save_for_backward(kept)
        ctx.n = x.shape[1]
        return ops.gather_pack(x.contiguous(), kept)

    @staticmethod
    def backward(ctx, g):
        (kept,) = ctx.saved_tensors
        return ops.scatter_unpack(g.contiguous(), kept, ctx.n), None


@mode_recorded
class HeadFn(torch.autograd.Function):
    """Final LayerNorm + classifier head on the CLS row.  Returns (logits [B,C], features = normed tokens[:, 1:])."""

    @staticmethod
    def forward(ctx, x, nw, nb, hw, hb, eps):
        B, n, D = x.shape
        M = B * n
        x = x.contiguous()
        train = wants_grad(ctx)
        xn, mean, rstd = ops.layernorm_fwd(x, ops.contiguous_map(M, D), nw, nb, M, D, eps, stats=train)
        C = hw.shape[0]
        logits = torch.empty((B, C), dtype=torch.float32, device=x.device)
        ops.gemm(ops.NT, xn, n * D, hw, D, logits, C, B, C, D, ops.EPI_BIAS, hb)   # A = CLS rows (row stride n*D)
        if train:
            ctx.save_for_backward(x, nw, hw, mean, rstd, xn, nb, hb)
            ctx.dims = (B, n, D, C)
        return logits, xn.view(B, n, D)[:, 1:]

    @staticmethod
    def backward(ctx, glogits, gfeat):
        x, nw, hw, mean, rstd, xn, nb, hb = ctx.saved_tensors
        B, n, D, C = ctx.dims
        M = B * n
        dev = x.device
        gfull = torch.zeros((B, n, D), dtype=torch.float32, device=dev)
        if gfeat is not None and n > 1:
            gfeat = gfeat.contiguous()
            ops.copy_rows(gfeat, ops.contiguous_map(B * (n - 1), D), B * (n - 1), D, dst=gfull, dst_map=ops.skip_cls_map(n, D))
        dhw = dhb = None
        if glogits is not None:
            glogits = glogits.contiguous()
            cls_g = ops.linear_dgrad(glogits, hw)
            ops.copy_rows(cls_g, ops.contiguous_map(B, D), B, D, dst=gfull, dst_map=(1, n * D, D, 0))
            cls_rows = ops.copy_rows(xn, (1, n * D, D, 0), B, D) if _need(ctx, 3) else None
            dhw, dhb = ops.linear_param_grads(glogits, cls_rows, hw, hb, _need(ctx, 3), _need(ctx, 4))
        gx = torch.empty((M, D), dtype=torch.float32, device=dev)
        dnw = ops.grad_buffer(nw) if (_need(ctx, 1) or _need(ctx, 2)) else None
        dnb = ops.grad_buffer(nb) if dnw is not None else None
        ops.layernorm_bwd(x, ops.contiguous_map(M, D), gfull.view(M, D), nw, mean, rstd, gx, None, dnw, dnb, M, D)
        return (gx.view(B, n, D) if _need(ctx, 0) else None, dnw if _need(ctx, 1) else None, dnb if _need(ctx, 2) else None,
                dhw, dhb, None)


def as_policy(policy, B, n):
    """[B,n,1] / [B,n] keep policy of the reference (:892-894) -> contiguous fp32 [B,n] (no copy when it already is)."""
    p = policy.reshape(B, n)
    if p.dtype != torch.float32:
        p = p.float()
    return p.contiguous()


def ragged_block_forward(xp, cu, B, max_n, params, heads, eps, scale, want_cls=False):
    """One transformer block on a ragged packed batch [total, D] (inference with a dynamic keep ratio, :935-949: every image keeps its
    own number of tokens).  LayerNorm and the four GEMMs run over all packed rows at once, attention per image through cu_seqlens.
    Forward only.  Returns (y [total, D], cls_rows [H, total] or None)."""
    n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = params
    total, D = xp.shape
    cmap = ops.contiguous_map(total, D)
    ln1, _, _ = ops.layernorm_fwd(xp, cmap, n1w, n1b, total, D, eps, stats=False)
    qkv = ops.linear_fwd(ln1, qkvw, qkvb)
    ao, cls_rows = ops.attn_varlen_fwd(qkv, cu, B, total, max_n, heads, scale, want_cls=want_cls)
    x1 = ops.linear_fwd(ao, projw, projb, epi=ops.EPI_BIAS_RESID, aux=xp)
    ln2, _, _ = ops.layernorm_fwd(x1, cmap, n2w, n2b, total, D, eps, stats=False)
    h = ops.linear_fwd(ln2, fc1w, fc1b, epi=ops.EPI_BIAS_GELU)
    return ops.linear_fwd(h, fc2w, fc2b, epi=ops.EPI_BIAS_RESID, aux=x1), cls_rows


def rows_map_3d(t):
    """Row map (relative to t.data_ptr()) of a [B, R, C] tensor whose last dim is dense."""
    assert t.dim() == 3 and t.stride(2) == 1
    return (t.shape[1], t.stride(0), t.stride(1), 0)


class RowLossFn(torch.autograd.Function):
    """mean-over-`denom` of a per-row loss from d2s_kl_rows: cross entropy, KL between two log-softmaxes, or KL against a
    probability target (losses.py:94-95,196,198-203,220-225).  Gradient flows to `s` only (the targets are teacher
    outputs).  `s` is [rows, C] contiguous or a [B, R, C] view with a dense last dim."""

    @staticmethod
    def forward(ctx, s, mode, t, t_ids, labels, denom, *extra):
        row_weight = extra[0] if extra else None      # optional 7th input: per-row weights [rows]
        ctx.nextra = len(extra)
        if s.dim() == 3:
            smap = rows_map_3d(s)
            rows, C = s.shape[0] * s.shape[1], s.shape[2]
        else:
            s = s.contiguous()
            rows, C = s.shape
            smap = ops.contiguous_map(rows, C)
        tmap = (1, 0, 0, 0)
        tid = None
        if t is not None:
            if t_ids is not None:       # teacher tokens gathered by kept ids: row r -> t[b, ids[b, j]]
                assert t.dim() == 3 and t.stride(2) == 1
                tmap = (t_ids.shape[1], t.stride(0), t.stride(1), 0)
                tid = t_ids.contiguous().view(-1)
            elif t.dim() == 3:
                tmap = rows_map_3d(t)
            else:
                t = t.contiguous()
                tmap = ops.contiguous_map(rows, C)
        loss_row, grad = ops.kl_rows(s, smap, rows, C, mode, t=t, t_map=tmap, t_ids=tid, labels=labels, want_grad=True,
                                     row_weight=row_weight)
        ctx.save_for_backward(grad)
        ctx.meta = (tuple(s.shape), float(denom))
        return ops.sum_scalar(loss_row, 1.0 / float(denom))

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        shape, denom = ctx.meta
        gs = ops.scale_by_scalar(grad, g.contiguous(), 1.0 / denom)
        return (gs.view(shape), None, None, None, None, None) + (None,) * ctx.nextra


class AddClsPosFn(torch.autograd.Function):
    """tokens [B,T,D] -> [B,T+1,D]: prepend the CLS token and add the position table (t2t_vit.py:160-162)."""

    @staticmethod
    def forward(ctx, tok, cls_token, pos_embed):
        B, T, D = tok.shape
        out = torch.empty((B, T + 1, D), dtype=torch.float32, device=tok.device)
        lib_call = ops.lib.call
        lib_call("d2s_assemble_tokens", ops.lib.ptr(tok.contiguous()), ops.lib.ptr(cls_token), ops.lib.ptr(pos_embed), ops.lib.ptr(out),
                 B, T, D)
        ctx.save_for_backward(cls_token, pos_embed)
        ctx.dims = (B, T, D)
        return out

    @staticmethod
    def backward(ctx, g):
        cls_token, pos_embed = ctx.saved_tensors
        B, T, D = ctx.dims
        g = g.contiguous()
        dtok = ops.copy_rows(g, ops.skip_cls_map(T + 1, D), B * T, D).view(B, T, D) if ctx.needs_input_grad[0] else None
        dcls = ops.batch_sum(g, ops.grad_buffer(cls_token), B, D, (T + 1) * D) if ctx.needs_input_grad[1] else None
        dpos = ops.batch_sum(g, ops.grad_buffer(pos_embed), B, (T + 1) * D, (T + 1) * D) if ctx.needs_input_grad[2] else None
        return dtok, dcls, dpos


class PolicySoftmaxFn(torch.autograd.Function):
    """Attention.softmax_with_policy (dynamic_vit.py:195-214): attn [B,H,N,N], policy [B,N,1] or [B,N] -> probabilities."""

    @staticmethod
    def forward(ctx, attn, policy, eps):
        B, H, N, _ = attn.shape
        attn = attn.contiguous()
        pol = policy.reshape(B, N).contiguous().float()
        out = torch.empty_like(attn)
        ops.lib.call("d2s_softmax_policy_fwd", ops.lib.ptr(attn), ops.lib.ptr(pol), ops.lib.ptr(out), B, H, N, float(eps))
        ctx.save_for_backward(attn, pol)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        attn, pol = ctx.saved_tensors
        B, H, N, _ = attn.shape
        ga = torch.empty_like(attn)
        ops.lib.call("d2s_softmax_policy_bwd", ops.lib.ptr(attn), ops.lib.ptr(pol), ops.lib.ptr(g.contiguous()), ops.lib.ptr(ga), B, H, N,
                     ctx.eps)
        return ga, None, None


def select_topk(keep_probs, k):
    """Hard top-k of the keep probabilities (dynamic_vit.py:858-862): (kept, dropped) int64, each ascending."""
    return ops.select_topk(keep_probs.detach().contiguous(), k)


@mode_recorded
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for the stand-alone Mlp / Attention modules (act: None | "gelu" | "relu")."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x.contiguous()
        z = None
        if act == "gelu":
            z = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
            y = ops.linear_fwd(x, w, b, epi=ops.EPI_BIAS_GELU, aux_out=z)
        elif act == "relu":
            y = ops.linear_fwd(x, w, b, epi=ops.EPI_BIAS_RELU)
            z = y
        else:
            y = ops.linear_fwd(x, w, b)
        ctx.act = act
        ctx.save_for_backward(x, w, z if z is not None else x)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, z = ctx.saved_tensors
        g = g.contiguous()
        if ctx.act == "gelu":     # dz = g * gelu'(z): identity-weight GEMMs are wasteful, so use the elementwise route
            g = ops.act_grad(g, z, "gelu")
        elif ctx.act == "relu":
            g = ops.act_grad(g, z, "relu")
        dx = ops.linear_dgrad(g, w) if ctx.needs_input_grad[0] else None
        db = torch.empty((w.shape[0],), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(g, x, torch.empty_like(w), db=db)
        elif db is not None:
            ops.colsum(g, db)
        return dx, dw, db, None


@mode_recorded
class AttnCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v on the raw qkv Linear output [B*n, 3*H*64] (+ CLS softmax row); with a 7th input `policy` [B, n]
    the softmax is Attention.softmax_with_policy (:195-214), fused."""

    @staticmethod
    def forward(ctx, qkv, B, n, H, scale, want_cls, *extra):
        qkv = qkv.contiguous()
        policy = extra[0] if extra else None
        ctx.nextra = len(extra)
        if policy is None:
            out, lse, cls_row = ops.attn_fwd(qkv, B, n, H, scale, want_cls)
            ctx.save_for_backward(qkv, out, lse)
        else:
            out, lse, cinv, cls_row = ops.attn_policy_fwd(qkv, policy, B, n, H, scale, want_cls=want_cls)
            ctx.save_for_backward(qkv, out, lse, cinv, policy)
        ctx.dims = (B, n, H, scale)
        if cls_row is None:
            cls_row = torch.empty((0,), device=qkv.device)
        ctx.mark_non_differentiable(cls_row)
        return out, cls_row

    @staticmethod
    def backward(ctx, g, _gc):
        B, n, H, scale = ctx.dims
        if len(ctx.saved_tensors) == 3:
            qkv, out, lse = ctx.saved_tensors
            dqkv = ops.attn_bwd(qkv, out, g.contiguous(), lse, B, n, H, scale)
        else:
            qkv, out, lse, cinv, policy = ctx.saved_tensors
            dqkv = ops.attn_policy_bwd(qkv, policy, out, g.contiguous(), lse, cinv, B, n, H, scale)
        return (dqkv, None, None, None, None, None) + (None,) * ctx.nextra


class PerturbedTopKFn(torch.autograd.Function):
    """vit_models/peturbed_topk.py:16-80 with the noise tensor as an explicit input."""

    @staticmethod
    def forward(ctx, x, noise, k, sigma):
        x, noise = x.contiguous(), noise.contiguous()
        ind = ops.perturbed_topk_fwd(x, noise, k, sigma)
        ctx.save_for_backward(x, noise)
        ctx.meta = (k, float(sigma))
        return ind

    @staticmethod
    def backward(ctx, g):
        x, noise = ctx.saved_tensors
        k, sigma = ctx.meta
        return ops.perturbed_topk_bwd(x, noise, g.contiguous(), k, sigma), None, None, None
