"""Counterpart of the reference's evaluate.py::evaluate_performance (evaluate.py:8-84), including its second ("unpruned")
forward pass, which at this commit of the reference is a duplicate of the first (model.unpruned is never read,
dynamic_vit.py:962).  evaluate_timing (evaluate.py:87-179) reads torch.cuda.Event attributes that are commented out in the
reference's model, i.e. it is dead code there; per-kernel timing on this path comes from rocprofv3 / bench.py instead."""
import torch

from d2s import functional as DF
from d2s import ops
from losses import MaskLoss


def evaluate_performance(args, model, teacher_model, val_data_loader):
    running_loss, running_acc, running_unpruned_acc, n = 0.0, 0.0, 0.0, 0
    model.eval()
    teacher_model.eval()
    mask_loss_fn = MaskLoss(args, "val")
    metrics = {}
    thr = getattr(args, "patch_score_threshold", None) is not None
    keep_ratio_batches = []
    with torch.no_grad():
        for val_inputs, val_labels in val_data_loader:
            val_inputs = val_inputs.to(args.device, non_blocking=True)
            val_labels = val_labels.to(args.device, non_blocking=True)
            cls_attn_weights = teacher_model.forward_cls_attention(val_inputs)          # :33
            outputs = model(val_inputs)                                                 # :35
            model.unpruned = True
            unpruned_logits, _, _, _ = model(val_inputs)                                # :37
            model.unpruned = False
            unpruned_preds = torch.argmax(unpruned_logits, dim=1)
            running_unpruned_acc += float((unpruned_preds == val_labels).sum()) / val_labels.shape[0]
            logits, cls_attns, pred_logits, kept_token_idx = outputs
            mask_loss_fn(pred_logits, cls_attn_weights, kept_token_idx, metrics)        # :44
            loss = DF.RowLossFn.apply(logits, ops.CE_LABEL, None, None, val_labels.contiguous(), logits.shape[0])   # :46
            preds = torch.argmax(logits, dim=1)
            running_loss += float(loss)
            running_acc += float((preds == val_labels).sum()) / val_labels.shape[0]
            n += 1
            if thr and model.keep_ratios is not None:                                   # :53-57
                keep_ratio_batches.append(model.keep_ratios)
    n = max(n, 1)
    if thr and keep_ratio_batches:                                                      # :59-62
        from utils import keep_ratio_summary
        (metrics["val_min_keep_ratio"], metrics["val_avg_keep_ratio"],
         metrics["val_max_keep_ratio"]) = keep_ratio_summary(keep_ratio_batches)
    metrics["val_loss"] = running_loss / n
    metrics["val_acc"] = running_acc / n
    metrics["unpruned_acc"] = running_unpruned_acc / n
    args.epoch_acc = metrics["val_acc"]
    print(f'val loss: {metrics["val_loss"]:.4f}, acc: {metrics["val_acc"]:.4f}')
    return metrics
