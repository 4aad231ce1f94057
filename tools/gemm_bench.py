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
B = int(os.environ.get("D2S_BENCH_B", "128"))
shapes = []
for n, tag in ((197, "teacher/student n=197"), (99, "student n=99")):
    M = B * n
    shapes += [("NT", M, 1152, 384, tag + " qkv"), ("NT", M, 384, 384, tag + " proj"), ("NT", M, 1536, 384, tag + " fc1"),
               ("NT", M, 384, 1536, tag + " fc2")]
    if n == 99 or True:
        shapes += [("NN", M, 1536, 384, tag + " d(fc2)"), ("NN", M, 384, 1536, tag + " d(fc1)"), ("NN", M, 384, 384, tag + " d(proj)"),
                   ("NN", M, 384, 1152, tag + " d(qkv)"),
                   ("TN", 384, 1536, M, tag + " wgrad fc2"), ("TN", 1536, 384, M, tag + " wgrad fc1"), ("TN", 384, 384, M, tag + " wgrad proj"),
                   ("TN", 1152, 384, M, tag + " wgrad qkv")]
Mp = B * 196
shapes += [("NT", Mp, 1536, 384, "pred in_conv"), ("NT", Mp, 768, 1536, "pred l0"), ("NT", Mp, 384, 768, "pred l1"),
           ("NT", Mp, 192, 384, "pred l2"), ("NT", Mp, 96, 192, "pred l3"), ("NT", Mp, 1, 96, "pred l4"),
           ("NT", Mp, 384, 768, "patch embed"), ("NT", 128, 1000, 384, "head")]
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
    for _ in range(3):
        ops.gemm(lay[L], A, lda, Bm, ldb, C, N, M, N, K, epi, bias)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    s.record()
    for _ in range(it):
        ops.gemm(lay[L], A, lda, Bm, ldb, C, N, M, N, K, epi, bias)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / it
    print(f"{L:6} {M:6d} {N:5d} {K:6d}  {us:8.1f} {2.0*M*N*K/us/1e6:7.1f}  {what}")
