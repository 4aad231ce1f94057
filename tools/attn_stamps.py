"""Per-workgroup phase timeline of the fp32 attention forward (GPU box only; needs `make -C dense2sparse-vit_amd/csrc attn-stamps`).
Stamps (shader cycles, s_memtime): 0 entry, 1 first K/V tile in LDS, 2 / 3 second / third tile in LDS, 4 key loop done, 5 exit; 8 / 9 = entry / exit
on the 100 MHz wall clock (s_memrealtime)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["D2S_LIB_PATH"] = os.path.join(REPO, "dense2sparse-vit_amd", "lib_diag", "libd2s_hip_attnstamps.so")
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import numpy as np
import torch
from d2s import ops, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B, H = 128, 6
dev = torch.device("cuda:0")
qkv = torch.randn(B * n, 3 * H * 64, device=dev)
for _ in range(3):
    ops.attn_fwd(qkv, B, n, H, 0.125)
torch.cuda.synchronize()
ops.attn_fwd(qkv, B, n, H, 0.125)
torch.cuda.synchronize()
nwg = min(8192, ((n + 127) // 128) * B * H)
buf = (ctypes.c_ulonglong * (16 * nwg))()
f = lib.load().d2s_debug_read_attn_stamps
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf, nwg) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 16).astype(np.int64)
t0 = s[:, 8].min()
ent = (s[:, 8] - t0) / 100.0          # us
ext = (s[:, 9] - t0) / 100.0
print(f"n {n}: {nwg} workgroups, kernel span {ext.max():.1f} us; entry times: p10 {np.percentile(ent,10):.1f} p50 {np.percentile(ent,50):.1f} p90 {np.percentile(ent,90):.1f} max {ent.max():.1f} us")
life = ext - ent
print(f"workgroup life: mean {life.mean():.1f} us, p10 {np.percentile(life,10):.1f}, p90 {np.percentile(life,90):.1f}")
cyc = lambda a, b: (s[:, b] - s[:, a]).astype(np.float64)
for name, a, b in (("entry -> tile 0 in LDS (Q, K0, V0 loads)", 0, 1), ("tile 0 compute + tile 1 staged", 1, 2), ("tile 1 compute + tile 2 staged", 2, 3),
                   ("tiles 2.. to end of key loop", 3, 4), ("epilogue (normalise, stores, lse, cls row)", 4, 5), ("whole", 0, 5)):
    d = cyc(a, b)
    print(f"  {name:48s} mean {d.mean():9.0f} cyc   p10 {np.percentile(d,10):9.0f}   p90 {np.percentile(d,90):9.0f}")
first = ent < 1.0
print(f"first-round workgroups ({first.sum()}): prologue mean {cyc(0,1)[first].mean():.0f} cyc; later ones ({(~first).sum()}): {cyc(0,1)[~first].mean() if (~first).any() else float('nan'):.0f} cyc")
