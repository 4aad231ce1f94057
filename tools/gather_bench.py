"""Per-launch duration of the kept-token gather at the headline shape, launches back to back (GPU box only).
Input buffers are rotated through > 512 MB so that reads come from HBM, not from the 256 MB Infinity Cache."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops

dev = torch.device("cuda:0")
B, n, k, D = 128, 197, 98, 384
nbuf = 16
xs = [torch.randn(B, n, D, device=dev) for _ in range(nbuf)]
ids = torch.stack([torch.randperm(n - 1, device=dev)[:k].sort().values for _ in range(B)])
alg = B * (2.0 * (k + 1) * D * 4 + 8.0 * k)
for variant in ["-"]:
    for _ in range(3):
        ops.gather_pack(xs[0], ids)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 200
    s.record()
    for i in range(it):
        ops.gather_pack(xs[i % nbuf], ids)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / it
    ref = torch.cat([xs[0][:, :1], torch.gather(xs[0][:, 1:], 1, ids[:, :, None].expand(-1, -1, D))], 1)
    ok = torch.equal(ops.gather_pack(xs[0], ids), ref)
    print(f"variant {variant}: {us:7.2f} us/launch (back to back, incl. launch gaps)  {alg / us / 1e3:7.1f} GB/s  bit-exact={ok}")

gs = [torch.randn(B, k + 1, D, device=dev) for _ in range(nbuf)]
alg_s = B * ((k + 1) * D * 4.0 + n * D * 4.0 + 8.0 * k)
for variant in ["-"]:
    for _ in range(3):
        ops.scatter_unpack(gs[0], ids, n)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(200):
        ops.scatter_unpack(gs[i % nbuf], ids, n)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 200
    ref = torch.zeros(B, n, D, device=dev)
    ref[:, :1] = gs[0][:, :1]
    ref[:, 1:].scatter_(1, ids[:, :, None].expand(-1, -1, D), gs[0][:, 1:])
    ok = torch.equal(ops.scatter_unpack(gs[0], ids, n), ref)
    print(f"scatter variant {variant}: {us:7.2f} us/launch back to back  {alg_s / us / 1e3:7.1f} GB/s  bit-exact={ok}")
