"""The `--predictor-bn` variant of the mask predictor (vit_models/dynamic_vit.py:438-476): the large predictor with every LayerNorm
replaced by BatchNormLayer (:350-367, BatchNorm1d over all B * N token rows).  Same GEMM / half-mean-concat / softmax kernels as
d2s.functional.PredictorFn; the normalisation is d2s_batchnorm_fwd / bwd with batch statistics in training mode (running estimates
updated in place) and the running estimates in eval mode.  Statistics are per process, exactly as in the reference (no cross-rank
synchronisation: SURVEY 8e caveat i)."""
import torch

from . import ops
from .functional import mode_recorded


@mode_recorded
class PredictorBNFn(torch.autograd.Function):
    """forward(x [B, n, D], training, n_layers running_mean..., running_var..., params...) -> (scores [B, n-1], keep_probs [B, n-1]).
    params: in_bn_w, in_bn_b, in_fc_w, in_fc_b, then 5 x (bn_w, bn_b, fc_w, fc_b); running: 6 x (mean, var) buffers."""

    @staticmethod
    def forward(ctx, x, training, running, *params):
        B, n, D = x.shape
        T = n - 1
        M = B * T
        x = x.contiguous()
        xt = ops.copy_rows(x, ops.skip_cls_map(n, D), M, D)                                  # x[:, 1:] as rows (:855)
        h0, mean0, rstd0 = ops.batchnorm_fwd(xt, params[0], params[1], running[0], running[1], training)
        a1 = ops.linear_fwd(h0, params[2], params[3], epi=ops.EPI_BIAS_RELU)
        C = a1.shape[1]
        cur = ops.half_mean_concat(a1, B, T, C)                                               # :540-544
        saved = [xt, h0, mean0, rstd0, a1]
        nl = (len(params) - 4) // 4
        for j in range(nl):
            bw, bb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            y, mean, rstd = ops.batchnorm_fwd(cur, bw, bb, running[2 + 2 * j], running[3 + 2 * j], training)
            last = j == nl - 1
            nxt = ops.linear_fwd(y, fw, fb, epi=ops.EPI_BIAS if last else ops.EPI_BIAS_RELU)
            saved += [cur, y, mean, rstd]
            cur = nxt
        scores = cur.view(B, T)
        probs = ops.softmax_rows(scores)                                                      # :551
        ctx.save_for_backward(*saved, *params)
        ctx.meta = (B, n, D, T, M, C, nl, len(saved), bool(training))
        ctx.mark_non_differentiable(probs)
        return scores, probs

    @staticmethod
    def backward(ctx, gscores, _gprobs):
        B, n, D, T, M, C, nl, nsaved, training = ctx.meta
        saved = ctx.saved_tensors[:nsaved]
        params = ctx.saved_tensors[nsaved:]
        xt, h0, mean0, rstd0, a1 = saved[:5]
        dev = gscores.device
        grads = [None] * len(params)
        want = [ctx.needs_input_grad[3 + i] for i in range(len(params))]
        d = gscores.contiguous().view(M, 1)
        for j in reversed(range(nl)):
            cur, y, mean, rstd = saved[5 + 4 * j: 9 + 4 * j]
            bw, bb, fw, fb = params[4 + 4 * j: 8 + 4 * j]
            base = 4 + 4 * j
            grads[base + 2], grads[base + 3] = ops.linear_param_grads(d, y, fw, fb, want[base + 2], want[base + 3])
            dy = ops.linear_dgrad(d, fw)
            dbw = ops.grad_buffer(bw) if (want[base] or want[base + 1]) else None
            dbb = ops.grad_buffer(bb) if dbw is not None else None
            # cur is the ReLU output of layer j - 1 for j >= 1: that ReLU's backward is folded into this layer's dx
            d = ops.batchnorm_bwd(cur, dy, bw, mean, rstd, dbw, dbb, training, relu_mask=(j >= 1))
            grads[base], grads[base + 1] = (dbw if want[base] else None), (dbb if want[base + 1] else None)
        dz1 = ops.half_mean_concat(d, B, T, C, relu_mask_src=a1)
        grads[2], grads[3] = ops.linear_param_grads(dz1, h0, params[2], params[3], want[2], want[3])
        gx = None
        if ctx.needs_input_grad[0] or want[0] or want[1]:
            dh0 = ops.linear_dgrad(dz1, params[2])
            dbw = ops.grad_buffer(params[0]) if (want[0] or want[1]) else None
            dbb = ops.grad_buffer(params[1]) if dbw is not None else None
            dxt = ops.batchnorm_bwd(xt, dh0, params[0], mean0, rstd0, dbw, dbb, training)
            grads[0], grads[1] = (dbw if want[0] else None), (dbb if want[1] else None)
            if ctx.needs_input_grad[0]:
                gx = torch.zeros((B, n, D), dtype=torch.float32, device=dev)
                ops.copy_rows(dxt, ops.contiguous_map(M, D), M, D, dst=gx, dst_map=ops.skip_cls_map(n, D))
        return (gx, None, None) + tuple(grads)
