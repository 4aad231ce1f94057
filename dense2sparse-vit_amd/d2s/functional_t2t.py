"""Autograd Functions of the Tokens-to-Token front end.  Same rules as d2s.functional: explicit
forward / backward sequences of C-ABI calls, no torch arithmetic.

Reference lines (relative to /root/reference/vit_models):
  UnfoldFn            t2t_vit.py:55-57,85,90-92,97-99   (nn.Unfold + transpose, and the token -> image re-structurisation)
  TokenPerformerFn    token_performer.py:45-59           (dropout p is treated as 0: parity is defined for eval mode)
  TokenTransformerFn  token_transformer.py:26-60
  LayerNormFn         t2t_vit.py:166-168                 (per-block norm of forward_features)
"""
import torch

from . import ops
from .functional import mode_recorded


def _need(ctx, i):
    return ctx.needs_input_grad[i]


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shape = x.shape
        D = shape[-1]
        x2 = x.contiguous().view(-1, D)
        M = x2.shape[0]
        y, mean, rstd = ops.layernorm_fwd(x2, ops.contiguous_map(M, D), w, b, M, D, eps)
        ctx.save_for_backward(x2, w, b, mean, rstd)
        ctx.shape = shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, g):
        x2, w, b, mean, rstd = ctx.saved_tensors
        M, D = x2.shape
        dx = torch.empty_like(x2)
        dw = ops.grad_buffer(w) if (_need(ctx, 1) or _need(ctx, 2)) else None
        db = ops.grad_buffer(b) if dw is not None else None
        ops.layernorm_bwd(x2, ops.contiguous_map(M, D), g.contiguous().view(M, D), w, mean, rstd, dx, None, dw, db, M, D)
        return dx.view(ctx.shape), (dw if _need(ctx, 1) else None), (db if _need(ctx, 2) else None), None


class UnfoldFn(torch.autograd.Function):
    """Soft split.  `x` is either an image [B, C, H, W] or a token tensor [B, H*W, C] (hw given) that the reference would first
    transpose/reshape to an image; it is read in place through strides."""

    @staticmethod
    def forward(ctx, x, k, s, p, hw):
        x = x.contiguous()
        if x.dim() == 4:
            B, C, H, W = x.shape
            strides = (C * H * W, H * W, W, 1)
        else:
            B, HW, C = x.shape
            H = W = hw
            assert H * W == HW
            strides = (HW * C, 1, W * C, C)
        ctx.meta = (tuple(x.shape), strides, B, C, H, W, k, s, p)
        return ops.unfold_fwd(x, strides, B, C, H, W, k, s, p)

    @staticmethod
    def backward(ctx, g):
        shape, strides, B, C, H, W, k, s, p = ctx.meta
        if not _need(ctx, 0):
            return None, None, None, None, None
        dsrc = torch.empty(shape, dtype=torch.float32, device=g.device)
        ops.unfold_bwd(g.contiguous(), dsrc, strides, B, C, H, W, k, s, p)
        return dsrc, None, None, None, None


def _mlp_tail_fwd(y, n2w, n2b, f1w, f1b, f2w, f2b, eps):
    """x = y + fc2(gelu(fc1(LN(y)))) on [M, E] rows; returns (out, saved)."""
    M, E = y.shape
    h2, mean2, rstd2 = ops.layernorm_fwd(y, ops.contiguous_map(M, E), n2w, n2b, M, E, eps)
    z = torch.empty((M, f1w.shape[0]), dtype=torch.float32, device=y.device)
    g_ = ops.linear_fwd(h2, f1w, f1b, epi=ops.EPI_BIAS_GELU, aux_out=z)
    out = ops.linear_fwd(g_, f2w, f2b, epi=ops.EPI_BIAS_RESID, aux=y)
    return out, (h2, mean2, rstd2, z, g_)


def _mlp_tail_bwd(gout, y, saved, n2w, n2b, f1w, f1b, f2w, f2b, want):
    """Returns (gy = gradient w.r.t. y incl. the residual path, [dn2w, dn2b, df1w, df1b, df2w, df2b])."""
    h2, mean2, rstd2, z, g_ = saved
    M, E = y.shape
    grads = [None] * 6
    grads[4], grads[5] = ops.linear_param_grads(gout, g_, f2w, f2b, want[4], want[5])
    dz = ops.linear_dgrad(gout, f2w, epi=ops.EPI_MUL_GELU_GRAD, aux=z)
    grads[2], grads[3] = ops.linear_param_grads(dz, h2, f1w, f1b, want[2], want[3])
    dh2 = ops.linear_dgrad(dz, f1w)
    gy = torch.empty((M, E), dtype=torch.float32, device=gout.device)
    dw = ops.grad_buffer(n2w) if (want[0] or want[1]) else None
    db = ops.grad_buffer(n2b) if dw is not None else None
    ops.layernorm_bwd(y, ops.contiguous_map(M, E), dh2, n2w, mean2, rstd2, gy, gout, dw, db, M, E)
    grads[0], grads[1] = (dw if want[0] else None), (db if want[1] else None)
    return gy, grads


@mode_recorded
class TokenPerformerFn(torch.autograd.Function):
    """Token_performer.forward on [B, T, dim]: LN -> kqv -> FAVOR+ attention -> v + proj -> LN -> MLP + residual."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, kqvw, kqvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, wfeat):
        B, T, dim = x.shape
        M = B * T
        x2 = x.contiguous().view(M, dim)
        eps = 1e-5
        h, mean1, rstd1 = ops.layernorm_fwd(x2, ops.contiguous_map(M, dim), n1w, n1b, M, dim, eps)
        kqv = ops.linear_fwd(h, kqvw, kqvb)
        ya, kp, qp, A, ksum, D = ops.performer_attn_fwd(kqv, wfeat, B, T)
        y = torch.empty((M, 64), dtype=torch.float32, device=x.device)
        ops.gemm(ops.NT, ya, 64, projw, 64, y, 64, M, 64, 64, ops.EPI_BIAS_RESID, projb, kqv[:, 128:], 192)   # y = v + proj(ya)
        out, tail = _mlp_tail_fwd(y, n2w, n2b, f1w, f1b, f2w, f2b, eps)
        ctx.save_for_backward(x2, n1w, n1b, kqvw, kqvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, wfeat, h, mean1, rstd1, kqv, ya, kp,
                              qp, A, ksum, D, y, *tail)
        ctx.dims = (B, T, dim)
        return out.view(B, T, 64)

    @staticmethod
    def backward(ctx, gout):
        (x2, n1w, n1b, kqvw, kqvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, wfeat, h, mean1, rstd1, kqv, ya, kp, qp, A, ksum, D, y,
         *tail) = ctx.saved_tensors
        B, T, dim = ctx.dims
        M = B * T
        want = [_need(ctx, i) for i in range(14)]
        gout = gout.contiguous().view(M, 64)
        gy, tg = _mlp_tail_bwd(gout, y, tail, n2w, n2b, f1w, f1b, f2w, f2b, want[7:13])
        grads = [None] * 14
        grads[7:13] = tg
        grads[5], grads[6] = ops.linear_param_grads(gy, ya, projw, projb, want[5], want[6])
        dya = ops.linear_dgrad(gy, projw)
        dkqv = ops.performer_attn_bwd(kqv, wfeat, ya, kp, qp, A, ksum, D, dya, gy, B, T)
        grads[3], grads[4] = ops.linear_param_grads(dkqv, h, kqvw, kqvb, want[3], want[4])
        if want[0] or want[1] or want[2]:
            dh = ops.linear_dgrad(dkqv, kqvw)
            dx = torch.empty((M, dim), dtype=torch.float32, device=gout.device)
            dw = ops.grad_buffer(n1w) if (want[1] or want[2]) else None
            db = ops.grad_buffer(n1b) if dw is not None else None
            ops.layernorm_bwd(x2, ops.contiguous_map(M, dim), dh, n1w, mean1, rstd1, dx, None, dw, db, M, dim)
            grads[0] = dx.view(B, T, dim) if want[0] else None
            grads[1], grads[2] = (dw if want[1] else None), (db if want[2] else None)
        return tuple(grads)


@mode_recorded
class TokenTransformerFn(torch.autograd.Function):
    """Token_transformer.forward on [B, T, dim]: LN -> qkv (1 head of width 64, scale dim**-0.5) -> fused attention ->
    v + proj -> LN -> MLP + residual."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b):
        B, T, dim = x.shape
        M = B * T
        x2 = x.contiguous().view(M, dim)
        eps = 1e-5
        scale = float(dim) ** -0.5
        h, mean1, rstd1 = ops.layernorm_fwd(x2, ops.contiguous_map(M, dim), n1w, n1b, M, dim, eps)
        qkv = ops.linear_fwd(h, qkvw, qkvb)
        ao, lse, _ = ops.attn_fwd(qkv, B, T, 1, scale, want_cls=False)
        y = torch.empty((M, 64), dtype=torch.float32, device=x.device)
        ops.gemm(ops.NT, ao, 64, projw, 64, y, 64, M, 64, 64, ops.EPI_BIAS_RESID, projb, qkv[:, 128:], 192)      # y = v + proj(attn)
        out, tail = _mlp_tail_fwd(y, n2w, n2b, f1w, f1b, f2w, f2b, eps)
        ctx.save_for_backward(x2, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, h, mean1, rstd1, qkv, ao, lse, y, *tail)
        ctx.dims = (B, T, dim, scale)
        return out.view(B, T, 64)

    @staticmethod
    def backward(ctx, gout):
        (x2, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, h, mean1, rstd1, qkv, ao, lse, y, *tail) = ctx.saved_tensors
        B, T, dim, scale = ctx.dims
        M = B * T
        want = [_need(ctx, i) for i in range(13)]
        gout = gout.contiguous().view(M, 64)
        gy, tg = _mlp_tail_bwd(gout, y, tail, n2w, n2b, f1w, f1b, f2w, f2b, want[7:13])
        grads = [None] * 13
        grads[7:13] = tg
        grads[5], grads[6] = ops.linear_param_grads(gy, ao, projw, projb, want[5], want[6])
        dao = ops.linear_dgrad(gy, projw)
        dqkv = ops.attn_bwd(qkv, ao, dao, lse, B, T, 1, scale)
        # skip connection through v: dqkv[:, 128:192] += gy
        ops.copy_rows(gy, ops.contiguous_map(M, 64), M, 64, dst=dqkv, dst_map=(M, 0, 192, 128), accumulate=True)
        grads[3], grads[4] = ops.linear_param_grads(dqkv, h, qkvw, qkvb, want[3], want[4])
        if want[0] or want[1] or want[2]:
            dh = ops.linear_dgrad(dqkv, qkvw)
            dx = torch.empty((M, dim), dtype=torch.float32, device=gout.device)
            dw = ops.grad_buffer(n1w) if (want[1] or want[2]) else None
            db = ops.grad_buffer(n1b) if dw is not None else None
            ops.layernorm_bwd(x2, ops.contiguous_map(M, dim), dh, n1w, mean1, rstd1, dx, None, dw, db, M, dim)
            grads[0] = dx.view(B, T, dim) if want[0] else None
            grads[1], grads[2] = (dw if want[1] else None), (db if want[2] else None)
        return tuple(grads)
