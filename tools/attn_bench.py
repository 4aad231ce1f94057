"""Fused attention forward / backward per-launch time at the model's shapes (GPU box only): B=128, H=6, dh=64."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
dev = torch.device("cuda:0")
B, H = 128, 6
for n in (197, 99):
    qkv = torch.randn(B * n, 3 * H * 64, device=dev)
    dout = torch.randn(B * n, H * 64, device=dev)
    out, lse, _ = ops.attn_fwd(qkv, B, n, H, 0.125)
    def fwd(): ops.attn_fwd(qkv, B, n, H, 0.125)
    def bwd(): ops.attn_bwd(qkv, out, dout, lse, B, n, H, 0.125)
    for name, fn, fl in (("fwd", fwd, 4.0 * n * n * 64 * B * H), ("bwd", bwd, 10.0 * n * n * 64 * B * H)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 30
        print(f"n {n:4d} {name} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (algorithmic, unpadded)")

print("bf16 arithmetic mode (forward on the bf16 matrix cores):")
ops.set_gemm_mode(ops.GEMM_BF16)
for n in (197, 99, 577):
    Bn = B if n != 577 else 64
    Hn = H if n != 577 else 12
    qkv = torch.randn(Bn * n, 3 * Hn * 64, device=dev)
    for _ in range(3): ops.attn_fwd(qkv, Bn, n, Hn, 0.125)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.attn_fwd(qkv, Bn, n, Hn, 0.125)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 30
    print(f"n {n:4d} B {Bn} H {Hn} fwd bf16 {us:8.1f} us  {4.0 * n * n * 64 * Bn * Hn / us / 1e6:7.1f} TF/s (algorithmic, unpadded)")
    out, lse, _ = ops.attn_fwd(qkv, Bn, n, Hn, 0.125)
    dout = torch.randn(Bn * n, Hn * 64, device=dev)
    for _ in range(3): ops.attn_bwd(qkv, out, dout, lse, Bn, n, Hn, 0.125)
    torch.cuda.synchronize()
    s.record()
    for _ in range(20): ops.attn_bwd(qkv, out, dout, lse, Bn, n, Hn, 0.125)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 20
    print(f"n {n:4d} B {Bn} H {Hn} bwd bf16 {us:8.1f} us  {10.0 * n * n * 64 * Bn * Hn / us / 1e6:7.1f} TF/s")
ops.set_gemm_mode(ops.GEMM_EXACT)
for n in (577,):
    qkv = torch.randn(64 * n, 3 * 12 * 64, device=dev)
    for _ in range(3): ops.attn_fwd(qkv, 64, n, 12, 0.125)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.attn_fwd(qkv, 64, n, 12, 0.125)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 10
    print(f"n {n:4d} B 64 H 12 fwd fp32 {us:8.1f} us  {4.0 * n * n * 64 * 64 * 12 / us / 1e6:7.1f} TF/s")
