import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd")); sys.path.insert(0, REPO)
import torch, importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
student, teacher = bench.build(dev, 0.5)
targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
ts = TrainStep(student, teacher, targs, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (128,), device=dev)
for _ in range(3): ts(x, y)
torch.cuda.synchronize()
hs = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts(x, y); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0, t2 - t0))
print("host enqueue per step %.1f ms (min %.1f), step wall %.1f ms" % (1e3 * sum(h[0] for h in hs) / 10, 1e3 * min(h[0] for h in hs), 1e3 * sum(h[1] for h in hs) / 10))
