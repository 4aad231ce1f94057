"""Vendor library (torch.matmul -> hipBLASLt) on the headline's fp32 WEIGHT-GRADIENT shapes dW = dy^T x beside this repository's TN kernel
(whole call: split-K slabs + ordered combine + bias column sums).  Yardstick only.  GPU box only."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps
print(f"{'tokens':>6s} {'n_out':>5s} {'n_in':>5s}   library us  TF/s    this repo us  TF/s   (dW = dy^T x, fp32; this repo incl. db)")
for tokens in (25216, 12672):
    for n_out, n_in in ((1152, 384), (384, 384), (1536, 384), (384, 1536)):
        dy = torch.randn(tokens, n_out, device=dev); x = torch.randn(tokens, n_in, device=dev)
        dW = torch.empty(n_out, n_in, device=dev); db = torch.empty(n_out, device=dev)
        dyt = dy.t()
        t_lib = timeit(lambda: torch.matmul(dyt, x, out=dW))
        t_our = timeit(lambda: ops.linear_wgrad(dy, x, dW, db=db))
        fl = 2.0 * tokens * n_out * n_in
        print(f"{tokens:6d} {n_out:5d} {n_in:5d}   {t_lib:9.1f} {fl / t_lib / 1e6:6.1f}    {t_our:9.1f} {fl / t_our / 1e6:6.1f}")
