"""LayerNorm forward/backward per-launch time at the model's shapes (GPU box only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
dev = torch.device("cuda:0")
for rows, D in [(128 * 197, 384), (128 * 99, 384), (128 * 196, 1536), (128 * 196, 768)]:
    x = torch.randn(rows, D, device=dev); dy = torch.randn(rows, D, device=dev); add = torch.randn(rows, D, device=dev)
    w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    cmap = ops.contiguous_map(rows, D)
    y, mean, rstd = ops.layernorm_fwd(x, cmap, w, b, rows, D, 1e-6)
    dx = torch.empty_like(x); dw = torch.empty(D, device=dev); db = torch.empty(D, device=dev)
    def fwd(): ops.layernorm_fwd(x, cmap, w, b, rows, D, 1e-6)
    def bwd(): ops.layernorm_bwd(x, cmap, dy, w, mean, rstd, dx, add, dw, db, rows, D)
    for name, fn, nbytes in (("fwd", fwd, 2 * rows * D * 4), ("bwd(+fold)", bwd, 4 * rows * D * 4)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 50
        print(f"rows {rows:6d} D {D:5d} {name:10s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
