"""Hot-path subset of the reference's utils.py: the optimiser parameter groups and the per-epoch learning-rate / freeze
schedule, with the reference's signatures (utils.py:67-147).  (The reference file itself needs torchvision and the whole model
zoo to import.)"""
import math

import torch


def get_param_groups(model, args):
    """utils.py:67-90."""
    decay, no_decay, predictor, early_exit = [], [], [], []
    for name, param in model.named_parameters():
        if 'predictor' in name or 'dist' in name:
            predictor.append(param)
        elif 'early_exit' in name:
            early_exit.append(param)
        elif not param.requires_grad:
            continue
        elif 'cls_token' in name or 'pos_embed' in name:
            continue
        elif len(param.shape) == 1 or name.endswith(".bias"):
            no_decay.append(param)
        else:
            decay.append(param)
    return [{'params': predictor, 'weight_decay': args.weight_decay, 'name': 'predictor'},
            {'params': no_decay, 'weight_decay': 0., 'name': 'base_no_decay'},
            {'params': decay, 'weight_decay': args.weight_decay, 'name': 'base_decay'},
            {'params': early_exit, 'weight_decay': args.weight_decay, 'name': 'early_exit'}]


def adjust_learning_rate(param_groups, args, step, model, warming_up_step=2, warmup_predictor=False, base_multi=0.1):
    """utils.py:93-147 for torch.optim param groups (d2s.engine.adjust_learning_rate is the same schedule for FusedAdamW)."""
    if getattr(args, "topk_selection", False):
        args.current_sigma = max(0, (1 - step / args.epochs) * args.initial_sigma)
    cos_lr = (math.cos(step / args.epochs * math.pi) + 1) * 0.5
    cos_lr = args.min_lr + cos_lr * (args.lr - args.min_lr)
    for n, p in model.named_parameters():
        p.requires_grad_(True if ('dist' in n or 'predictor' in n) else step >= args.warmup_steps)
    predictor_lr = cos_lr
    backbone_lr = 0 if step < args.warmup_steps else min(args.lr * 0.01, cos_lr)
    print(f'### Using lr {backbone_lr:.7f} for BACKBONE, cosine lr = {predictor_lr:.7f} for PREDICTOR')
    for param_group in param_groups:
        if param_group['name'] == 'predictor':
            param_group['lr'] = predictor_lr
            for p in param_group['params']:
                p.requires_grad_(predictor_lr != 0)
        elif param_group['name'] != 'early_exit':
            param_group['lr'] = backbone_lr
            for p in param_group['params']:
                p.requires_grad_(backbone_lr != 0)


class SyntheticLoader:
    """Deterministic stand-in for the ImageFolder loaders of build_data_sets.py: `steps` batches of N(0,1) images."""

    def __init__(self, steps, batch, img_size=224, num_classes=1000, seed=0, device="cpu", last_batch=None):
        """last_batch: size of the final batch (the reference's loaders use drop_last=False, ddp_training.py:15-20, so an epoch usually
        ends on a shorter batch); None = as long as the others."""
        self.steps, self.batch, self.img, self.nc, self.seed, self.device = steps, batch, img_size, num_classes, seed, device
        self.last_batch = last_batch

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed)
        for i in range(self.steps):
            b = self.last_batch if (self.last_batch and i == self.steps - 1) else self.batch
            yield (torch.randn((b, 3, self.img, self.img), generator=g, device=self.device),
                   torch.randint(0, self.nc, (b,), generator=g, device=self.device))


def parse_args(argv=None):
    """The reference's command line (utils.py:182-317): same flag names, types and defaults, so job scripts written for the
    reference run unchanged.  Data-set / augmentation / tracking flags are accepted and ignored (inputs are synthetic here:
    no data set or network in the image); four synthetic-run flags are added at the end."""
    import argparse
    p = argparse.ArgumentParser(description='Transformers')
    p.add_argument('--arch', default='deit_small', type=str)
    p.add_argument('--is-sbatch', action='store_true', default=False)
    p.add_argument('--wandb', action='store_true', default=False)
    p.add_argument('--save-path', default='test_imgs/')
    p.add_argument('--model-name', type=str, default='deit_small_patch16_224')
    p.add_argument('--patch-size', default=16)
    p.add_argument('--use-shape', action='store_true', default=False)
    p.add_argument('--batch-size', default=64, type=int)
    p.add_argument('--epochs', default=25, type=int)
    p.add_argument('--use-dp', action='store_true', default=False)
    p.add_argument('--use-ddp', action='store_true', default=False)
    p.add_argument('--imgnet-val-dir', type=str, default="")
    p.add_argument('--weight-decay', type=float, default=0.05)
    p.add_argument('--lr', type=float, default=5e-4)
    p.add_argument('--warmup-lr', type=float, default=1e-6)
    p.add_argument('--min-lr', type=float, default=1e-5)
    p.add_argument('--warmup-steps', default=5, type=int)
    p.add_argument('--early-exit', action='store_true', default=False)
    p.add_argument('--pruning-locs', nargs='+', default=[3], type=int)
    p.add_argument('--keep-ratios', nargs='+', type=float, default=[0.3])
    p.add_argument('--softmax-temp', default=1.0, type=float)
    p.add_argument('--use-ratio-loss', action='store_true', default=False)
    p.add_argument('--ratio-weight', default=2.0, type=float)
    p.add_argument('--use-token-dist-loss', action='store_true', default=False)
    p.add_argument('--dist-weight', default=0.5, type=float)
    p.add_argument('--teacher-cls-loss', action='store_true', default=False)
    p.add_argument('--cls-weight', default=1.0, type=float)
    p.add_argument('--topk-selection', action='store_true', default=False)
    p.add_argument('--mean-heads', action='store_true', default=False)
    p.add_argument('--random-drop', action='store_true', default=False)
    p.add_argument('--initial-sigma', default=0.05, type=float)
    p.add_argument('--attn-selection', action='store_true', default=False)
    p.add_argument('--cls-from-teacher', action='store_true', default=False)
    p.add_argument('--freeze-backbone', action='store_true', default=False)
    p.add_argument('--visualize-patch-drop', action='store_true', default=False)
    p.add_argument('--visualize-cls-attn-evo', action='store_true', default=False)
    p.add_argument('--small-predictor', action='store_true', default=False)
    p.add_argument('--mask-loss-type', default='kl_div', type=str)
    p.add_argument('--predictor-bn', action='store_true', default=False)
    p.add_argument('--patch-score-threshold', default=None, type=float)
    p.add_argument('--color-jitter', type=float, default=0.4)
    p.add_argument('--aa', type=str, default='rand-m9-mstd0.5-inc1')
    p.add_argument('--smoothing', type=float, default=0.1)
    p.add_argument('--train-interpolation', type=str, default='bicubic')
    p.add_argument('--repeated-aug', action='store_true')
    p.add_argument('--no-repeated-aug', action='store_false', dest='repeated_aug')
    p.set_defaults(repeated_aug=True)
    p.add_argument('--reprob', type=float, default=0.25)
    p.add_argument('--remode', type=str, default='pixel')
    p.add_argument('--recount', type=int, default=1)
    p.add_argument('--resplit', action='store_true', default=False)
    p.add_argument('--mixup', type=float, default=0.8)
    p.add_argument('--cutmix', type=float, default=1.0)
    p.add_argument('--cutmix-minmax', type=float, nargs='+', default=None)
    p.add_argument('--mixup-prob', type=float, default=1.0)
    p.add_argument('--mixup-switch-prob', type=float, default=0.5)
    p.add_argument('--mixup-mode', type=str, default='batch')
    # ---- additions for synthetic / offline runs (not in the reference) ----
    p.add_argument('--steps-per-epoch', type=int, default=20, help='synthetic training batches per epoch')
    p.add_argument('--val-steps', type=int, default=2, help='synthetic validation batches per epoch')
    p.add_argument('--student-checkpoint', type=str, default=None, help='local DeiT checkpoint for the student (weights_only load)')
    p.add_argument('--teacher-checkpoint', type=str, default=None, help='local DeiT checkpoint for the teacher (weights_only load)')
    p.add_argument('--torch-optim', action='store_true', default=False,
                   help="the reference's recipe (torch.optim.AdamW over get_param_groups) instead of the fused arena step")
    p.add_argument('--gemm-mode', choices=['exact', 'split', 'bf16'], default='exact')
    return p.parse_args(argv)


def keep_ratio_summary(keep_ratio_batches):
    """min / avg / max of the per-image keep ratios collected over an epoch (train.py:67-70,77-80; evaluate.py:53-62).  The loaders use
    drop_last=False (ddp_training.py:15-20), so the last batch may be shorter than the others: the batches are concatenated, never
    stacked.  avg follows the reference: the mean of the per-batch means (sum(avg_keep_ratio) / len(loader)), min / max over all images."""
    import torch
    allr = torch.cat([r.reshape(-1) for r in keep_ratio_batches])
    avg = torch.stack([r.float().mean() for r in keep_ratio_batches]).mean()
    return float(allr.min()), float(avg), float(allr.max())
