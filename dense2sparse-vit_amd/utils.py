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

    def __init__(self, steps, batch, img_size=224, num_classes=1000, seed=0, device="cpu"):
        self.steps, self.batch, self.img, self.nc, self.seed, self.device = steps, batch, img_size, num_classes, seed, device

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed)
        for _ in range(self.steps):
            yield (torch.randn((self.batch, 3, self.img, self.img), generator=g, device=self.device),
                   torch.randint(0, self.nc, (self.batch,), generator=g, device=self.device))
