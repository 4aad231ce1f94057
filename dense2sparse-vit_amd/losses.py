"""Counterpart of the reference's losses.py: same classes, constructor arguments, forward signatures and metrics keys
(including the swapped train_token_kl_loss / train_cls_kl_loss keys of losses.py:238-239), computed by HIP kernels.

Differences that are deliberate and documented:
  * running statistics stay on the device (no .item() per step, losses.py:111,230-233 force a D2H sync every step);
    metrics values are 0-d tensors, `float(v)` reads them.
  * the kl_div and mse mask losses, hard-label and soft-target (mixup) cross entropy are on the accelerated path; the reference's bce
    branch is broken as written (undefined `args` / `self.mask_criterions`, losses.py:57-58).
  * dynamic keep ratio (args.patch_score_threshold set): the reference's losses cannot run on that path (MaskLoss iterates over the
    batch dimension of the mask tensor it is handed, losses.py:81; BackboneLoss indexes with a float row and an undefined `C`,
    :216-218).  The fix built here keeps their intent: no token is removed in training, so every stage's KL runs over all N tokens
    against the un-gathered teacher target and its accuracy compares the two THRESHOLD masks (student scores vs teacher target under
    the same cumulative rule); the token distillation term is the mean over the tokens the last stage's mask keeps.
"""
import torch

from d2s import functional as DF
from d2s import ops


class MaskLoss(torch.nn.Module):
    def __init__(self, args, phase):
        super().__init__()
        self.phase = phase
        self.keep_ratios = args.keep_ratios
        self.loss_type = args.mask_loss_type
        if self.loss_type not in ("kl_div", "mse"):
            raise NotImplementedError("mask_loss_type 'kl_div' (losses.py:75-96) and 'mse' (:61-73) are on the accelerated path; the "
                                      "reference's 'bce' branch is broken as written (:57-58)")
        self.patch_score_threshold = getattr(args, "patch_score_threshold", None)
        self.count = 1
        self.running_loss = 0
        self.runnings_accs = [0 for _ in self.keep_ratios]

    def _forward_threshold(self, pred_logits, target, keep_masks, mask_accs):
        """Dynamic keep ratio: every stage scores all N tokens (nothing was gathered), see the module docstring."""
        B, T = target.shape
        mask_loss = 0
        with torch.no_grad():
            gt_mask, _ = ops.select_threshold(target, float(self.patch_score_threshold))
        for i in range(len(keep_masks)):
            if self.loss_type == "mse":
                mask_loss = mask_loss + DF.RowLossFn.apply(pred_logits[i], ops.MSE_TARGET, target, None, None, B * T / 100.0)
                continue
            with torch.no_grad():
                mask_accs[i] = ops.sum_scalar(ops.dense_mask_agreement(keep_masks[i].contiguous(), gt_mask), 1.0 / float(B * T))
            mask_loss = mask_loss + DF.RowLossFn.apply(pred_logits[i], ops.KL_PROB_TARGET, target, None, None, B)
        return mask_loss

    def forward(self, pred_logits, cls_attn_weights, kept_token_idx, metrics, accumulate=True):
        """accumulate=False: only the loss of this batch is computed (its running mean / accuracies are NOT advanced); the caller hands
        `self.last` to accumulate() later - the split a captured (hipGraph) step needs: the kernels replay, the host-side running
        statistics of losses.py:105-119 are advanced once per replay (d2s.engine.TrainStep)."""
        target = ops.teacher_target(cls_attn_weights.contiguous())             # losses.py:76-79
        B = target.shape[0]
        mask_loss = 0
        mask_accs = [0 for _ in self.keep_ratios]
        if self.patch_score_threshold is not None:
            mask_loss = self._forward_threshold(pred_logits, target, kept_token_idx, mask_accs)
            kept_token_idx = []
        for i in range(len(kept_token_idx)):
            if i > 0:
                ratio = self.keep_ratios[i] / self.keep_ratios[i - 1]
                gt_vals = ops.gather_renorm(target, kept_token_idx[i - 1], normalize=False)   # :84-85
                target = ops.gather_renorm(target, kept_token_idx[i - 1], normalize=True)     # :89-90
            else:
                ratio = self.keep_ratios[i]
                gt_vals = target
            T = gt_vals.shape[1]
            nk = int(T * ratio)                                                               # :132,154
            if self.loss_type == "mse":      # :61-73: 100 * mean squared difference between the raw scores and the target; no accuracy
                mask_loss = mask_loss + DF.RowLossFn.apply(pred_logits[i], ops.MSE_TARGET, target, None, None, B * T / 100.0)
                continue
            with torch.no_grad():
                gt_ids, _ = ops.select_topk(gt_vals, nk)
                pm_ids, _ = ops.select_topk(ops.softmax_rows(pred_logits[i].detach().contiguous()), nk)
                agree = ops.mask_agreement(pm_ids, gt_ids, T)
                mask_accs[i] = ops.sum_scalar(agree, 1.0 / float(B * T))             # :96
            mask_loss = mask_loss + DF.RowLossFn.apply(pred_logits[i], ops.KL_PROB_TARGET, target, None, None, B)   # :94-95
        self.last = (mask_loss.detach(), mask_accs)
        if accumulate:
            self.accumulate(metrics, self.last)
        return mask_loss

    def accumulate(self, metrics, last):
        """Advance the running mean of the loss and of the per-stage mask accuracies by one batch (losses.py:105-119)."""
        mask_loss, mask_accs = last
        self.running_loss = self.running_loss + mask_loss
        metrics[f"{self.phase}_mask_loss"] = self.running_loss / self.count
        for i, _ in enumerate(self.keep_ratios):
            self.runnings_accs[i] = self.runnings_accs[i] + mask_accs[i]
            metrics[f"{self.phase}_mask_acc_{i}"] = self.runnings_accs[i] / self.count
        self.count += 1


class BackboneLoss(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.soft_targets = getattr(args, "mixup", 0.) > 0.      # losses.py:170-174: SoftTargetCrossEntropy under mixup, else hard labels
        self.patch_score_threshold = getattr(args, "patch_score_threshold", None)
        self.count = 1
        self.running_loss = 0
        self.running_cls_loss = 0
        self.running_token_kl_loss = 0
        self.running_token_dist_loss = 0
        self.runnings_acc = 0

    def forward(self, logits_s, token_s, logits_t, token_t, kept_token_idx, train_labels, metrics, accumulate=True):
        """accumulate: see MaskLoss.forward."""
        B = logits_s.shape[0]
        if self.soft_targets:      # train_labels are [B, classes] probabilities produced by the caller's mixup_fn (train.py:29-30)
            cls_loss = DF.RowLossFn.apply(logits_s, ops.SOFT_CE, train_labels.float().contiguous(), None, None, B)
        else:
            cls_loss = DF.RowLossFn.apply(logits_s, ops.CE_LABEL, None, None, train_labels.contiguous(), B)      # :196
        cls_kl_loss = DF.RowLossFn.apply(logits_s, ops.KL_LOGIT_TARGET, logits_t.detach(), None, None, B)       # :198-203
        rows = token_s.shape[0] * token_s.shape[1]
        if self.patch_score_threshold is not None:
            # the intent of losses.py:216-225: distil only the tokens the last stage keeps; student and teacher both still hold all N
            # tokens, so the pairing is positional and 'batchmean' becomes the mean over the kept rows (weights = mask / count)
            w = ops.mask_row_weights(kept_token_idx[-1].detach().contiguous())
            token_kl_loss = DF.RowLossFn.apply(token_s, ops.KL_LOGIT_TARGET, token_t.detach(), None, None, 1.0, w)
        else:
            # teacher tokens gathered with the LAST stage's stage-relative ids, exactly like losses.py:212
            token_kl_loss = DF.RowLossFn.apply(token_s, ops.KL_LOGIT_TARGET, token_t.detach(), kept_token_idx[-1], None, rows)   # :218-225
        backbone_loss = cls_loss + cls_kl_loss + token_kl_loss
        self.last = (backbone_loss.detach(), cls_loss.detach(), cls_kl_loss.detach(), token_kl_loss.detach())
        self.last_terms = self.last[1:]
        if accumulate:
            self.accumulate(metrics, self.last)
        return backbone_loss

    def accumulate(self, metrics, last):
        """Advance the running means by one batch (losses.py:228-241)."""
        backbone_loss, cls_loss, cls_kl_loss, token_kl_loss = last
        self.running_loss = self.running_loss + backbone_loss
        self.running_cls_loss = self.running_cls_loss + cls_loss
        self.running_token_dist_loss = self.running_token_dist_loss + cls_kl_loss
        self.running_token_kl_loss = self.running_token_kl_loss + token_kl_loss
        metrics["train_backbone_loss"] = self.running_loss / self.count
        metrics["train_cls_loss"] = self.running_cls_loss / self.count
        metrics["train_token_kl_loss"] = self.running_token_dist_loss / self.count      # sic: swapped in the reference
        metrics["train_cls_kl_loss"] = self.running_token_kl_loss / self.count          # (losses.py:238-239)
        self.count += 1
