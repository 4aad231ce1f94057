#!/usr/bin/env python3
"""Data-parallel correctness check: N ranks (each with batch B/N) must produce, after the bucketed all-reduce, the same
averaged gradients and the same updated parameters as one process with the concatenated batch B.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ddp_check.py

Backend: RCCL ("nccl") when every rank has its own GPU, otherwise gloo on GPU tensors (rehearsal on a 1-GPU box: all ranks
share cuda:0).  Exit code 0 = pass.
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "dense2sparse-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from tests import cases


def build(case, dev):
    import vit_models
    cfg = case["cfg"]
    common = dict(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                  num_heads=cfg["heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True, num_classes=cfg["num_classes"])
    s = vit_models.VisionTransformerDiffPruning(pruning_loc=list(cfg["pruning_loc"]), token_ratio=list(cfg["token_ratio"]),
                                                distill=True, topk_selection=True, predictor_loss_type="kl_div", **common)
    t = vit_models.VisionTransformerTeacher(**common)
    sd_s, sd_t = cases.make_weights(case)
    s.load_state_dict({k: torch.from_numpy(v) for k, v in sd_s.items()})
    t.load_state_dict({k: torch.from_numpy(v) for k, v in sd_t.items()})
    args = types.SimpleNamespace(keep_ratios=list(cfg["token_ratio"]), mask_loss_type="kl_div", mixup=0.0,
                                 patch_score_threshold=None, step=0)
    return s.to(dev), t.to(dev), args


def main():
    from d2s.engine import TrainStep
    from d2s import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    own_gpu = ngpu >= world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if own_gpu else 0)
    torch.cuda.set_device(dev)
    if os.environ.get("D2S_FORCE_NCCL") == "1":      # rehearsal: RCCL with every rank on the same GPU (if the library allows it)
        own_gpu = True
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if own_gpu else "gloo", rank=rank, world_size=world)
    case = dict(cases.MODEL_CASES["micro2"])
    per = 2
    B = per * world
    x = torch.from_numpy(synth.images(B, 3, case["cfg"]["img_size"], seed=77))
    y = torch.from_numpy(synth.labels(B, case["cfg"]["num_classes"], seed=77))
    # tiny bucket so that several all-reduces are launched from inside backward (exercises the hook path); warmup_steps=1: epoch 0
    # trains the predictors only (live gradient set = their slices, exchanged by finish()), epoch 1 everything (hooks + buckets)
    s, t, args = build(case, dev)
    ts = TrainStep(s, t, args, distributed=True, bucket_mb=0.25, warmup_steps=1)
    ref = None
    if rank == 0:
        s1, t1, args1 = build(case, dev)
        ref = TrainStep(s1, t1, args1, distributed=False, warmup_steps=1)
    ok = True
    for epoch in (0, 1):
        ts.set_epoch(epoch)
        live = list(ts.reducer.live)
        if epoch == 0:
            assert ts.reducer.live_elems() < ts.arena.total // 2, "warm-up epoch must only exchange the predictor slices"
        info = ts(x[rank * per:(rank + 1) * per].to(dev), y[rank * per:(rank + 1) * per].to(dev))
        torch.cuda.synchronize()
        grads = ts.arena.grads.clone() / world
        params = ts.arena.params.clone()
        if rank == 0:
            ref.set_epoch(epoch)
            ref(x.to(dev), y.to(dev))
            torch.cuda.synchronize()
            sel = torch.cat([torch.arange(a, b) for a, b in live]).to(dev)       # the gradients the optimiser reads
            gd = float((grads[sel] - ref.arena.grads[sel]).norm() / ref.arena.grads[sel].norm())
            pd = float((params - ref.arena.params).abs().max())
            print(f"[ddp_check] world={world} backend={'nccl' if own_gpu else 'gloo'} epoch {epoch} live {len(sel)}/{ts.arena.total} "
                  f"rel grad diff {gd:.3e}  max param diff {pd:.3e}")
            ok = ok and gd < 1e-4 and pd < 2 * 2 * 5e-4 * 1.01
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag.to(dev) if own_gpu else flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
