"""Which framework-level ops issue device-to-device copies inside one training step (GPU box only)."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd")); sys.path.insert(0, REPO)
import torch
from torch.profiler import profile, ProfilerActivity
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
student, teacher = bench.build(dev, 0.5)
targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
ts = TrainStep(student, teacher, targs, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (128,), device=dev)
for _ in range(2): ts(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    ts(x, y); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if "copy" in e.key.lower() or "Memcpy" in e.key or "clone" in e.key.lower() or "contiguous" in e.key.lower() or "fill" in e.key.lower() or "zero" in e.key.lower() or "add" in e.key.lower()]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print(f"{e.key[:70]:70s} count {e.count:4d}  cuda_total {getattr(e, 'device_time_total', 0) / 1e3:8.3f} ms")
