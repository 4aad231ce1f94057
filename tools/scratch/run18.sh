#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python - <<'PY'
import os, sys, time, types
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "dense2sparse-vit_amd"))
import torch
import bench
from d2s import ops
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
for mode, batch in (("exact", 128), ("bf16", 128), ("exact", 32)):
    ops.set_gemm_mode({"exact": 0, "bf16": 2}[mode])
    student, teacher = bench.build(dev, 0.5)
    targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    ts = TrainStep(student, teacher, targs)
    x = torch.randn((batch, 3, 224, 224), device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    for _ in range(3): ts(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): ts(x, y)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"mode {mode} B={batch}: host enqueue {t_enq/10*1e3:.2f} ms/step, wall {t_all/10*1e3:.2f} ms/step")
    del ts, student, teacher
ops.set_gemm_mode(0)
PY
