"""One fp32 attention shape, a few launches - the target of a rocprofv3 --pmc pass (GPU box only).
usage: attn_one.py [n] [B] [H] [fwd|bwd]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 6
what = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dev = torch.device("cuda:0")
qkv = torch.randn(B * n, 3 * H * 64, device=dev)
out, lse, _ = ops.attn_fwd(qkv, B, n, H, 0.125)
dout = torch.randn(B * n, H * 64, device=dev)
for _ in range(5):
    if what == "fwd":
        ops.attn_fwd(qkv, B, n, H, 0.125)
    else:
        ops.attn_bwd(qkv, out, dout, lse, B, n, H, 0.125)
torch.cuda.synchronize()
