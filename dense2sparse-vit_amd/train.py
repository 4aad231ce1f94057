"""Counterpart of the reference's train.py: `train_one_epoch` with the same signature, call order, warm-up switch and
metrics keys (train.py:9-85).  The reference file cannot be imported as written (it imports fvcore and references an
un-imported `attention_segmentation`), so this is the step definition restated on the accelerated path.

`optimizer` may be a d2s.engine.TrainStep (flat arenas + fused AdamW + bucketed all-reduce: the fast path) or any
torch.optim optimizer built from utils.get_param_groups (the reference's recipe, mask_predictor.py:213-230)."""
import torch

from losses import MaskLoss, BackboneLoss


def train_one_epoch(args, model, teacher_model, train_data_loader, optimizer, mixup_fn=None):
    from d2s.engine import TrainStep
    if mixup_fn is not None and not getattr(args, "mixup", 0.) > 0.:
        raise ValueError("a mixup_fn needs args.mixup > 0 so that BackboneLoss uses the soft-target cross entropy (losses.py:170-172)")
    running_loss = 0.0
    metrics = {}
    model.train()
    teacher_model.eval()
    fast = isinstance(optimizer, TrainStep)
    if fast:
        step = optimizer
        step.metrics = metrics
        step.mask_loss_fn, step.backbone_loss_fn = MaskLoss(args, "train"), BackboneLoss(args)   # fresh running means per epoch (:14-15)
    else:
        mask_loss_fn, backbone_loss_fn = MaskLoss(args, "train"), BackboneLoss(args)
    n_steps = 0
    thr = getattr(args, "patch_score_threshold", None) is not None
    keep_ratio_batches = []      # :66-70 - kept on the device; the reference syncs three times per step for these statistics
    for train_step, train_data in enumerate(train_data_loader):
        train_inputs = train_data[0].to(args.device, non_blocking=True)
        train_labels = train_data[1].to(args.device, non_blocking=True)
        if mixup_fn is not None:
            train_inputs, train_labels = mixup_fn(train_inputs, train_labels)                        # :29-30 (the caller's transform)
        if fast:
            info = step(train_inputs, train_labels)
            mask_loss, train_loss = info["mask_loss"], info["loss"]
        else:
            with torch.no_grad():
                logits_t, token_t, cls_attn_weights = teacher_model(train_inputs)                    # :40
            logits_s, token_s, pred_logits, kept_token_idx = model(train_inputs)                     # :43
            mask_loss = mask_loss_fn(pred_logits, cls_attn_weights, kept_token_idx, metrics)          # :46
            backbone_loss = backbone_loss_fn(logits_s, token_s, logits_t, token_t, kept_token_idx, train_labels, metrics)   # :48
            train_loss = mask_loss if args.step < args.warmup_steps else backbone_loss + mask_loss   # :50-53
            optimizer.zero_grad()
            train_loss.backward()
            optimizer.step()
        if train_step % (400 if getattr(args, "is_sbatch", False) else 10) == 0:                     # :59-62 (one sync per 10 steps)
            print(f'training step_{train_step} mask loss: {float(mask_loss):.4f}, train loss: {float(train_loss):.4f}, ')
        running_loss = running_loss + train_loss.detach()
        n_steps += 1
        if thr and model.keep_ratios is not None:
            keep_ratio_batches.append(model.keep_ratios)
    if thr and keep_ratio_batches:                                                                   # :77-80 (the histogram plot is the caller's)
        from utils import keep_ratio_summary
        (metrics["train_min_keep_ratio"], metrics["train_avg_keep_ratio"],
         metrics["train_max_keep_ratio"]) = keep_ratio_summary(keep_ratio_batches)
    metrics["train_loss"] = float(running_loss) / max(n_steps, 1)                                    # :82
    print(f'train loss: {metrics["train_loss"]:.4f}')
    return metrics
