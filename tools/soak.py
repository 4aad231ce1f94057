"""Soak: N optimiser steps of the headline configuration back to back; prints throughput and the caching allocator's reserved bytes every 25 steps
(the host runs ahead of the GPU, so cross-stream frees are deferred: the pool must level off, not grow without bound).  GPU box only."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
import bench
from d2s import lib
from d2s.engine import TrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
lib.load()
student, teacher = bench.build(dev, 0.5)
targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
ts = TrainStep(student, teacher, targs, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0, distributed=False, graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn((128, 3, 224, 224), device=dev, generator=g)
labels = torch.randint(0, 1000, (128,), device=dev, generator=g)
for _ in range(5):
    ts(images, labels)
torch.cuda.synchronize()
t0 = time.perf_counter(); last = t0
for i in range(1, n + 1):
    info = ts(images, labels)
    if i % 25 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        st = torch.cuda.memory_stats()
        print(f"step {i:4d}: {25 * 128 / (now - last):7.0f} images/s  reserved {st['reserved_bytes.all.current'] / 2**30:6.2f} GiB  allocated {st['allocated_bytes.all.current'] / 2**30:6.2f} GiB  "
              f"device mallocs {st['num_device_alloc']}  loss {float(info['loss']):.5f}", flush=True)
        last = time.perf_counter()
