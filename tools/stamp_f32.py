#!/usr/bin/env python3
"""Diagnostic (debug build with -DD2S_STAMPS only): per-workgroup and per-CU timeline of the exact fp32 GEMM kernel."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops, lib
L = lib.load()
dev = torch.device("cuda:0")
for (M, N, K, bm, bn) in ((25216, 1152, 384, 128, 128), (25216, 1536, 384, 128, 64), (12672, 1536, 384, 64, 64)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
    for _ in range(10):
        ops.gemm(0, A, K, W, K, C, N, M, N, K, ops.EPI_BIAS, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.gemm(0, A, K, W, K, C, N, M, N, K, ops.EPI_BIAS, b); e.record(); torch.cuda.synchronize()
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    L.d2s_debug_read_stamps_f32.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.d2s_debug_read_stamps_f32(buf.ctypes.data, nwg) == 0
    st = buf.reshape(nwg, 8)
    t = st[:, :4].astype(np.float64); t0 = t[:, 0].min()
    ent, lb, le, ex = [(t[:, i] - t0) / 100.0 for i in range(4)]
    cyc = (st[:, 5] - st[:, 4]).astype(np.float64)
    hw = st[:, 6]
    hwid, xcc = (hw & 0xFFFFFFFF).astype(np.int64), (hw >> 32).astype(np.int64)
    cu = ((hwid >> 8) & 0xF) + 16 * ((hwid >> 13) & 0x7) + 128 * ((hwid >> 12) & 1) + 256 * (xcc & 0xF)
    print(f"M{M} N{N} K{K} tile {bm}x{bn}: launch {s.elapsed_time(e)*1000:.1f} us, {nwg} WGs, span {ex.max():.1f} us, distinct CU ids {len(np.unique(cu))}")
    print(f"   per WG median: prologue {np.median(lb-ent):.2f} us  loop {np.median(le-lb):.2f} us  epilogue {np.median(ex-le):.2f} us   loop clock {np.median(cyc/((le-lb)*100+1e-9))*100:.0f} MHz")
    mt, nt = bm // 64, bn // 64
    alone_cycles = (K // 2) * mt * nt * 64
    print(f"   MFMA cycles per WG-wave {alone_cycles}; loop cycles median {np.median(cyc):.0f} -> {np.median(cyc)/alone_cycles:.2f}x the MFMA-only time")
    # per-CU: number of WGs, busy span, and the sum of MFMA-only time the CU had to deliver
    occ, idle = [], []
    for c in np.unique(cu):
        m = cu == c
        span = ex[m].max() - ent[m].min()
        need = m.sum() * alone_cycles / (np.median(cyc / ((le - lb) * 100 + 1e-9)) * 100) # us of pure MFMA at the loop clock, one wave per SIMD per WG
        occ.append(m.sum()); idle.append(1 - need / span)
    print(f"   per CU: WGs min/med/max {min(occ)}/{int(np.median(occ))}/{max(occ)};  matrix-pipe idle fraction of the CU's busy span: median {np.median(idle):.2f}  min {min(idle):.2f}  max {max(idle):.2f}")
    order = np.argsort(ent)
    conc = []
    for c in np.unique(cu)[:64]:
        m = np.where(cu == c)[0]
        ev = sorted([(ent[i], 1) for i in m] + [(ex[i], -1) for i in m])
        cur = 0; last = ev[0][0]; acc = {}
        for tt, d in ev:
            acc[cur] = acc.get(cur, 0) + (tt - last); last = tt; cur += d
        tot = sum(acc.values())
        conc.append({k: v / tot for k, v in acc.items()})
    keys = sorted(set(k for d in conc for k in d))
    print("   time share by number of co-resident WGs on a CU:", {k: round(float(np.mean([d.get(k, 0) for d in conc])), 3) for k in keys})
