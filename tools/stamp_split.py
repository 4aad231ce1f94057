#!/usr/bin/env python3
"""Diagnostic (debug build with -DD2S_STAMPS only): where a workgroup of gemm_pieces_nt_kernel spends its life."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops, lib
L = lib.load()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops.set_gemm_mode(mode)
dev = torch.device("cuda:0")
for (M, N, K) in ((25216, 1536, 384), (25216, 384, 1536), (12672, 384, 384)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
    for _ in range(20):
        ops.gemm(0, A, K, W, K, C, N, M, N, K, ops.EPI_BIAS, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.gemm(0, A, K, W, K, C, N, M, N, K, ops.EPI_BIAS, b); e.record(); torch.cuda.synchronize()
    nwg = ((M + 127) // 128) * ((N + 127) // 128)
    buf = np.zeros(nwg * 16, dtype=np.uint64)
    L.d2s_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.d2s_debug_read_stamps(buf.ctypes.data, nwg) == 0
    st = buf.reshape(nwg, 16).astype(np.float64)
    t0 = st[:, 0].min()
    ent, lb, le, ex = [(st[:, i] - t0) / 100.0 for i in range(4)]
    cyc = st[:, 5] - st[:, 4]
    print(f"mode {mode} M{M} N{N} K{K}: launch {s.elapsed_time(e)*1000:.1f} us (incl. split passes); WGs {nwg}")
    print(f"   prologue (entry->loop) median {np.median(lb-ent):6.2f} us   loop median {np.median(le-lb):6.2f} us   epilogue median {np.median(ex-le):6.2f} us"
          f"   loop clock {np.median(cyc/((le-lb)*100+1e-9))*100:6.0f} MHz")
    print(f"   entry time p50/p100 {np.median(ent):6.1f}/{ent.max():6.1f} us   exit p50/p100 {np.median(ex):6.1f}/{ex.max():6.1f} us   kernel span {ex.max()-ent.min():6.1f} us")
    if K >= 6 * 32:
        ph = [np.median(st[:, i + 1] - st[:, i]) for i in range(8, 13)]
        print("   K-step 5, wave 0, cycles: LDS stores %.0f | barrier %.0f | issue next loads %.0f | ds_read+MFMA %.0f | barrier %.0f | total %.0f"
              % (ph[0], ph[1], ph[2], ph[3], ph[4], sum(ph)))
