"""Does the host run ahead of the GPU in the headline step, and if not, what blocks it?  Per step: host time inside TrainStep.__call__, the
caching allocator's device-malloc / free counters, and how far the GPU is behind when the call returns (GPU box only)."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
import bench
from d2s import ops, lib
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
lib.load()
student, teacher = bench.build(dev, 0.5)
targs = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
ts = TrainStep(student, teacher, targs, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0, distributed=False, graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn((128, 3, 224, 224), device=dev, generator=g)
labels = torch.randint(0, 1000, (128,), device=dev, generator=g)
for _ in range(4):
    ts(images, labels)
torch.cuda.synchronize()
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
prev = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
t_start = time.perf_counter()
for i in range(10):
    h0 = time.perf_counter()
    ts(images, labels)
    h1 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record()
    st = torch.cuda.memory_stats()
    d = {k: st.get(k, 0) - prev[k] for k in keys}; prev = {k: st.get(k, 0) for k in keys}
    print(f"step {i}: host {1e3 * (h1 - h0):6.2f} ms  since start {1e3 * (h1 - t_start):7.1f} ms  reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB  "
          f"active peak {st['active_bytes.all.peak'] / 2**30:.2f} GiB  {d}", flush=True)
torch.cuda.synchronize()
print(f"10 steps: {1e3 * (time.perf_counter() - t_start) / 10:.2f} ms per step wall")
# where inside the step does the host wait?  time the phases with the GPU left running
import d2s.engine as E
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
acc = {}
ts.forward_losses = timed("forward_losses", ts.forward_losses)
ts.opt.zero_grad = timed("zero_grad", ts.opt.zero_grad)
ts.opt.step = timed("opt.step", ts.opt.step)
ts.opt.refresh_transposed_weights = timed("refresh_wT", ts.opt.refresh_transposed_weights)
ts.arena.collect_grads = timed("collect_grads", ts.arena.collect_grads)
teacher_fwd = ts.teacher.forward; ts.teacher.forward = timed("teacher.forward", teacher_fwd)
student_fwd = ts.student.forward; ts.student.forward = timed("student.forward", student_fwd)
t0 = time.perf_counter()
for i in range(10):
    ts(images, labels)
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms per step by phase:", {k: round(1e2 * v, 2) for k, v in acc.items()}, "whole call", round(1e2 * tot, 2), "(backward = the rest)")
