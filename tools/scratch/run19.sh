#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python - <<'PY' > gpurun_out/x_hostprof.log 2>&1
import os, sys, time, types, cProfile, pstats, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "dense2sparse-vit_amd"))
import torch
import bench
from d2s import ops
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
student, teacher = bench.build(dev, 0.5)
targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
ts = TrainStep(student, teacher, targs)
x = torch.randn((32, 3, 224, 224), device=dev); y = torch.randint(0, 1000, (32,), device=dev)
for _ in range(3): ts(x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): ts(x, y)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35); print(s.getvalue()[:6000])
PY
head -70 gpurun_out/x_hostprof.log | cut -c1-160
