"""Counterpart of the reference's ddp_training.py intent (one process per GPU, DistributedSampler-style sharding, DDP gradient
averaging, per-epoch metric reduces + barrier; ddp_training.py:4-26,93,174-177,213).  The reference file has no imports and
calls APIs that no longer exist, so this is the working driver on the accelerated path.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ddp_training.py --epochs 2 --steps 10
"""
import argparse
import os
import types

import torch
import torch.distributed as dist


def setup(rank, world_size):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '12355')
    dist.init_process_group("nccl", rank=rank, world_size=world_size)      # "nccl" == RCCL over xGMI on ROCm


def cleanup():
    dist.destroy_process_group()


def train_model_ddp(rank, world_size, args):
    import vit_models
    from d2s.engine import TrainStep
    from train import train_one_epoch
    from utils import SyntheticLoader
    if world_size > 1:
        setup(rank, world_size)
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    args.device = torch.device("cuda", local)
    try:
        torch.manual_seed(0)
        student = vit_models.dynamic_vit_small_patch16_224_student(args.pruning_locs, args.keep_ratios, topk_selection=True,
                                                                   predictor_loss_type="kl_div").to(args.device)
        teacher = vit_models.dynamic_vit_small_patch16_224_teacher().to(args.device)
        step = TrainStep(student, teacher, args, lr=args.lr, min_lr=args.min_lr, weight_decay=args.weight_decay, epochs=args.epochs,
                         warmup_steps=args.warmup_steps, distributed=world_size > 1)
        if world_size > 1:
            dist.broadcast(step.arena.params, src=0)
        for epoch in range(args.epochs):
            step.set_epoch(epoch)
            loader = SyntheticLoader(args.steps, args.batch_size, seed=1000 * epoch + rank, device=args.device)   # per-rank shard
            metrics = train_one_epoch(args, student, teacher, loader, step)
            if world_size > 1:                                                     # ddp_training.py:174-177,213
                t = torch.tensor([metrics["train_loss"]], device=args.device)
                dist.reduce(t, 0, op=dist.ReduceOp.SUM)
                dist.barrier()
                if rank == 0:
                    print(f"epoch {epoch}: mean train loss over ranks {float(t) / world_size:.4f}")
    finally:
        if world_size > 1:
            cleanup()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--min-lr", type=float, default=1e-5)
    ap.add_argument("--weight-decay", type=float, default=0.05)
    ap.add_argument("--warmup-steps", type=int, default=0)
    ap.add_argument("--pruning-locs", type=int, nargs="+", default=[3])
    ap.add_argument("--keep-ratios", type=float, nargs="+", default=[0.5])
    a = ap.parse_args()
    a.mask_loss_type, a.mixup, a.patch_score_threshold, a.step, a.is_sbatch = "kl_div", 0.0, None, 0, False
    train_model_ddp(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), a)
