"""Counterpart of the reference's entry script (mask_predictor.py:56-330): the same flags (utils.parse_args), the same sequence
- student / teacher from the arch factories (:170-202), parameter groups + AdamW (:213-230), optional backbone freeze
(:218-224), per epoch adjust_learning_rate -> train_one_epoch -> evaluate_performance (:295-310), best-accuracy tracking -
on the accelerated path with synthetic batches (the image has no data set and no network; the reference's ImageNet folders,
mixup transforms, wandb / tensorboard tracking and visualisations are outside the path).

    python dense2sparse-vit_amd/mask_predictor.py --arch deit_small --pruning-locs 3 --keep-ratios 0.5 --epochs 3 \\
        --warmup-steps 1 --batch-size 64 --steps-per-epoch 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        dense2sparse-vit_amd/mask_predictor.py --use-ddp ...        (one process per GPU, RCCL)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import utils                                         # noqa: E402
import vit_models                                    # noqa: E402
from d2s import ops                                  # noqa: E402
from d2s.engine import TrainStep                     # noqa: E402
from evaluate import evaluate_performance            # noqa: E402
from train import train_one_epoch                    # noqa: E402

_STUDENTS = {"deit_tiny": "dynamic_vit_tiny_patch16_224_student", "deit_small": "dynamic_vit_small_patch16_224_student",
             "deit_base": "dynamic_vit_base_patch16_224_student"}
_TEACHERS = {"deit_tiny": "dynamic_vit_tiny_patch16_224_teacher", "deit_small": "dynamic_vit_small_patch16_224_teacher",
             "deit_base": "dynamic_vit_base_patch16_224_teacher"}


def check_supported(args):
    """Flags whose code path is outside the accelerated hot path fail here, loudly, instead of silently training something else."""
    bad = []
    if args.patch_score_threshold is not None:
        print("Attention: --patch-score-threshold: the reference's losses and inference branch cannot run on this path (losses.py:81,216-218, "
              "dynamic_vit.py:936); this build follows its training forward line by line and the documented fix for the rest (DESIGN.md section 10)")
        if len(args.pruning_locs) > 1:
            bad.append("--patch-score-threshold with more than one pruning stage (ragged inference supports one stage: the reference's "
                       "second stage cannot run, dynamic_vit.py:945-946)")
    if args.early_exit:
        print("Attention: --early-exit creates the extra head but, as in the reference, nothing calls it (dynamic_vit.py:752-758)")
    if args.random_drop:
        # stored on the model and used for the job name only (dynamic_vit.py:748, mask_predictor.py:79-80): no effect on the forward
        print("Attention: --random-drop has no effect on the forward pass at this commit of the reference (attribute only)")
    if args.mask_loss_type not in ("kl_div", "mse"):
        bad.append(f"--mask-loss-type {args.mask_loss_type} (kl_div and mse are on the path; bce is broken in the reference)")
    if args.use_dp:
        bad.append("--use-dp (one process per GPU only: --use-ddp under torch.distributed.run)")
    if bad:
        raise SystemExit("not on the accelerated path: " + "; ".join(bad))
    if args.predictor_bn and args.use_ddp:
        print("Attention: --predictor-bn keeps per-rank batch statistics (not synchronised), exactly like the reference")
    if args.mixup > 0 or args.cutmix > 0 or args.cutmix_minmax is not None:
        print("Attention: mixup/cutmix are not used (synthetic batches)")
    args.mixup, args.cutmix, args.cutmix_minmax = 0.0, 0.0, None


def build_models(args):
    """mask_predictor.py:170-202 (the arch switch), with local checkpoints instead of URL downloads."""
    arch = args.arch if args.arch in _STUDENTS else "deit_small"
    student = getattr(vit_models, _STUDENTS[arch])(args.pruning_locs, args.keep_ratios, topk_selection=args.topk_selection,
                                                   early_exit=args.early_exit, mean_heads=args.mean_heads,
                                                   random_drop=args.random_drop, small_predictor=args.small_predictor,
                                                   predictor_loss_type=args.mask_loss_type, predictor_bn=args.predictor_bn,
                                                   patch_score_threshold=args.patch_score_threshold,
                                                   checkpoint_path=args.student_checkpoint)
    teacher = getattr(vit_models, _TEACHERS[arch])(checkpoint_path=args.teacher_checkpoint)
    return student.to(args.device), teacher.to(args.device)


def main(argv=None):
    args = utils.parse_args(argv)
    check_supported(args)
    if not torch.cuda.is_available():
        raise SystemExit("mask_predictor.py needs a GPU: the path has no CPU fallback")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = args.use_ddp and world > 1
    torch.cuda.set_device(local)
    args.device = torch.device("cuda", local)
    args.world_size, args.job_name, args.nb_classes, args.step = world, "synthetic_job", 1000, 0
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)          # RCCL
    ops.set_gemm_mode({"exact": ops.GEMM_EXACT, "split": ops.GEMM_SPLIT, "bf16": ops.GEMM_BF16}[args.gemm_mode])
    torch.manual_seed(42)                                                     # mask_predictor.py:43-44
    student, teacher = build_models(args)
    if rank == 0:
        for key in sorted(vars(args), key=str.lower):
            print(f'{key}: {getattr(args, key)}')
    if args.freeze_backbone:                                                  # :218-224
        print('Freezing whole student, except predictor network')
        for n, p in student.named_parameters():
            p.requires_grad = 'predictor' in n
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    if args.torch_optim:
        if distributed:
            raise SystemExit("--torch-optim is the single-process reference recipe; drop it for --use-ddp")
        optim = torch.optim.AdamW(utils.get_param_groups(student, args), lr=args.lr, weight_decay=args.weight_decay)   # :213,229-230
    else:
        optim = TrainStep(student, teacher, args, lr=args.lr, min_lr=args.min_lr, weight_decay=args.weight_decay, epochs=args.epochs,
                          warmup_steps=args.warmup_steps, distributed=distributed)
        if distributed:
            dist.broadcast(optim.arena.params, src=0)
    n_pred = sum(p.numel() for n, p in student.named_parameters() if 'predictor' in n and p.requires_grad)
    print(f'Total number of trainable parameters in predictor network in millions: {n_pred / 1e6}')
    print(f"Start training for {args.epochs} epochs, with batch size of {args.batch_size}")
    img = 224
    since, best_acc = time.time(), 0.0
    for epoch in range(args.epochs):
        args.step = epoch
        print('Epoch {}/{}'.format(epoch + 1, args.epochs))
        print('-' * 50)
        if args.torch_optim:
            utils.adjust_learning_rate(optim.param_groups, args, epoch, student, warmup_predictor=False,
                                       warming_up_step=args.warmup_steps, base_multi=0.1)                  # :300-301
        else:
            optim.set_epoch(epoch)
        if args.topk_selection and hasattr(args, "current_sigma"):
            student.current_sigma = args.current_sigma
        train_loader = utils.SyntheticLoader(args.steps_per_epoch, args.batch_size, img, seed=1000 * epoch + rank, device=args.device)
        val_loader = utils.SyntheticLoader(args.val_steps, args.batch_size, img, seed=777 + rank, device=args.device)
        t0 = time.time()
        train_metrics = train_one_epoch(args, student, teacher, train_loader, optim, None)                  # :308
        torch.cuda.synchronize()
        dt = time.time() - t0
        val_metrics = evaluate_performance(args, student, teacher, val_loader)                              # :310
        epoch_metrics = dict(train_metrics, **val_metrics)
        if distributed:                                                                                     # ddp_training.py:174-177,213
            t = torch.tensor([epoch_metrics["train_loss"], epoch_metrics["val_acc"]], device=args.device)
            dist.all_reduce(t)
            epoch_metrics["train_loss"], epoch_metrics["val_acc"] = (t / world).tolist()
            dist.barrier()
        best_acc = max(best_acc, epoch_metrics['val_acc'])
        if rank == 0:
            print(f"epoch {epoch + 1}: {args.steps_per_epoch * args.batch_size * world / dt:.1f} train images/s, " +
                  ", ".join(f"{k}={v:.4f}" for k, v in sorted(epoch_metrics.items()) if isinstance(v, float)))
    elapsed = time.time() - since
    print(f'Training complete in {(elapsed // 60):.0f}m {(elapsed % 60):.0f}s')
    print(f'Best val acc: {best_acc:4f}')
    if distributed:
        dist.destroy_process_group()
    return best_acc


if __name__ == '__main__':
    main()
