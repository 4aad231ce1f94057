"""Small LayerNorm predictor (--small-predictor, vit_models/dynamic_vit.py:409-426): LN -> Linear(D,D) -> GELU, split / token
mean / concat, then LN -> Linear(D,D/2) -> GELU -> LN -> Linear(D/2,D/4) -> GELU -> LN -> Linear(D/4,1).  Same structure as
d2s.functional.PredictorFn with exact-erf GELU (pre-activations are saved, the GELU gradient is one elementwise kernel)."""
import torch

from . import ops
from .functional import mode_recorded


@mode_recorded
class SmallPredictorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        # params: in_ln_w, in_ln_b, in_fc_w, in_fc_b, then 3 x (ln_w, ln_b, fc_w, fc_b)
        B, n, D = x.shape
        T = n - 1
        M = B * T
        x = x.contiguous()
        eps = 1e-5
        h0, mean0, rstd0 = ops.layernorm_fwd(x, ops.skip_cls_map(n, D), params[0], params[1], M, D, eps)
        z1 = torch.empty((M, params[2].shape[0]), dtype=torch.float32, device=x.device)
        a1 = ops.linear_fwd(h0, params[2], params[3], epi=ops.EPI_BIAS_GELU, aux_out=z1)
        C = a1.shape[1]
        cur = ops.half_mean_concat(a1, B, T, C)
        saved = [x, h0, mean0, rstd0, z1]
        nl = (len(params) - 4) // 4
        for j in range(nl):
            lw, lb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            width = cur.shape[1]
            ln, mean, rstd = ops.layernorm_fwd(cur, ops.contiguous_map(M, width), lw, lb, M, width, eps)
            if j == nl - 1:
                nxt, z = ops.linear_fwd(ln, fw, fb), ln       # z slot unused for the last layer (placeholder keeps indexing regular)
            else:
                z = torch.empty((M, fw.shape[0]), dtype=torch.float32, device=x.device)
                nxt = ops.linear_fwd(ln, fw, fb, epi=ops.EPI_BIAS_GELU, aux_out=z)
            saved += [cur, ln, mean, rstd, z]
            cur = nxt
        scores = cur.view(B, T)
        probs = ops.softmax_rows(scores)
        ctx.save_for_backward(*saved, *params)
        ctx.meta = (B, n, D, T, M, C, nl, len(saved))
        ctx.mark_non_differentiable(probs)
        return scores, probs

    @staticmethod
    def backward(ctx, gscores, _gprobs):
        B, n, D, T, M, C, nl, nsaved = ctx.meta
        saved = ctx.saved_tensors[:nsaved]
        params = ctx.saved_tensors[nsaved:]
        x, h0, mean0, rstd0, z1 = saved[:5]
        dev = gscores.device
        grads = [None] * len(params)
        want = [ctx.needs_input_grad[1 + i] for i in range(len(params))]
        d = gscores.contiguous().view(M, 1)
        for j in reversed(range(nl)):
            cur, ln, mean, rstd, _z = saved[5 + 5 * j: 10 + 5 * j]
            lw, lb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            base = 4 + 4 * j
            width = cur.shape[1]
            grads[base + 2], grads[base + 3] = ops.linear_param_grads(d, ln, fw, fb, want[base + 2], want[base + 3])
            dln = ops.linear_dgrad(d, fw)
            dcur = torch.empty((M, width), dtype=torch.float32, device=dev)
            dlw = ops.grad_buffer(lw) if (want[base] or want[base + 1]) else None
            dlb = ops.grad_buffer(lb) if dlw is not None else None
            ops.layernorm_bwd(cur, ops.contiguous_map(M, width), dln, lw, mean, rstd, dcur, None, dlw, dlb, M, width)
            grads[base], grads[base + 1] = (dlw if want[base] else None), (dlb if want[base + 1] else None)
            if j >= 1:      # cur = gelu(z_{j-1}): gradient w.r.t. the previous layer's pre-activation
                d = ops.act_grad(dcur, saved[5 + 5 * (j - 1) + 4], "gelu")
            else:
                d = dcur
        dz1 = ops.act_grad(ops.half_mean_concat(d, B, T, C), z1, "gelu")
        grads[2], grads[3] = ops.linear_param_grads(dz1, h0, params[2], params[3], want[2], want[3])
        gx = None
        if ctx.needs_input_grad[0] or want[0] or want[1]:
            dh0 = ops.linear_dgrad(dz1, params[2])
            gx = torch.zeros((B, n, D), dtype=torch.float32, device=dev)
            dlw = ops.grad_buffer(params[0]) if (want[0] or want[1]) else None
            dlb = ops.grad_buffer(params[1]) if dlw is not None else None
            ops.layernorm_bwd(x, ops.skip_cls_map(n, D), dh0, params[0], mean0, rstd0, gx, None, dlw, dlb, M, D)
            grads[0], grads[1] = (dlw if want[0] else None), (dlb if want[1] else None)
            if not ctx.needs_input_grad[0]:
                gx = None
        return (gx,) + tuple(grads)
