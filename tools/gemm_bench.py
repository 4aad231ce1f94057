#!/usr/bin/env python3
"""Per-shape timing of d2s_gemm_f32 on the GEMM shapes of the DeiT-S keep-0.5 step (GPU box only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops

dev = torch.device("cuda:0")
from d2s import lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # 0 exact f32 MFMA, 1 bf16x3 split (fp32-class), 2 bf16
ops.set_gemm_mode(mode)
print("gemm mode", mode)
base = os.environ.get("D2S_BENCH_MODEL", "small") == "base"      # DeiT-B 384x384 keep 0.3 (config 5) instead of DeiT-S 224 keep 0.5
B = int(os.environ.get("D2S_BENCH_B", "64" if base else "128"))
D = 768 if base else 384
shapes = []
for n, tag in (((577, "teacher/student n=577"), (173, "student n=173")) if base else ((197, "teacher/student n=197"), (99, "student n=99"))):
    M = B * n
    shapes += [("NT", M, 3 * D, D, tag + " qkv"), ("NT", M, D, D, tag + " proj"), ("NT", M, 4 * D, D, tag + " fc1"),
               ("NT", M, D, 4 * D, tag + " fc2")]
    shapes += [("NN", M, 4 * D, D, tag + " d(fc2)"), ("NN", M, D, 4 * D, tag + " d(fc1)"), ("NN", M, D, D, tag + " d(proj)"),
               ("NN", M, D, 3 * D, tag + " d(qkv)"),
               ("TN", D, 4 * D, M, tag + " wgrad fc2"), ("TN", 4 * D, D, M, tag + " wgrad fc1"), ("TN", D, D, M, tag + " wgrad proj"),
               ("TN", 3 * D, D, M, tag + " wgrad qkv")]
Mp = B * (576 if base else 196)
shapes += [("NT", Mp, 4 * D, D, "pred in_conv"), ("NT", Mp, 2 * D, 4 * D, "pred l0"), ("NT", Mp, D, 2 * D, "pred l1"),
           ("NT", Mp, D // 2, D, "pred l2"), ("NT", Mp, D // 4, D // 2, "pred l3"), ("NT", Mp, 1, D // 4, "pred l4"),
           ("NT", Mp, D, 768, "patch embed"), ("NT", B, 1000, D, "head")]
only = os.environ.get("D2S_BENCH_ONLY")     # substring filter on "layout what", e.g. "NT teacher/student n=197 fc1"
if only:
    shapes = [t for t in shapes if only in f"{t[0]} {t[4]}"]
lay = {"NT": 0, "NN": 1, "TN": 2}
print(f"{'layout':6} {'M':>6} {'N':>5} {'K':>6}  {'us':>8} {'TF/s':>7}  what")
for L, M, N, K, what in shapes:
    if L == "NT":
        A = torch.randn(M, K, device=dev); Bm = torch.randn(N, K, device=dev); lda, ldb = K, K
    elif L == "NN":
        A = torch.randn(M, K, device=dev); Bm = torch.randn(K, N, device=dev); lda, ldb = K, N
    else:
        A = torch.randn(K, M, device=dev); Bm = torch.randn(K, N, device=dev); lda, ldb = M, N
    if os.environ.get("D2S_BENCH_ZEROS") == "1":      # power / clock probe: all-zero operands toggle no multiplier bits
        A.zero_(); Bm.zero_()
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    epi = ops.EPI_BIAS if L == "NT" else ops.EPI_NONE
    kw = {}
    io = os.environ.get("D2S_BENCH_IO", "0")      # bf16 mode: 1 = A given in bf16, 2 = also a bf16 copy of C, 3 = bf16 copy only (no fp32 C)
    if io != "0" and mode == 2 and L != "TN":
        kw["a16"] = A.bfloat16()
        if io in ("2", "3"):
            kw["c16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if io == "3":
            C = None
    for _ in range(3):
        ops.gemm(lay[L], A, lda, Bm, ldb, C, N, M, N, K, epi, bias, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    s.record()
    for _ in range(it):
        ops.gemm(lay[L], A, lda, Bm, ldb, C, N, M, N, K, epi, bias, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / it
    print(f"{L:6} {M:6d} {N:5d} {K:6d}  {us:8.1f} {2.0*M*N*K/us/1e6:7.1f}  {what}")
