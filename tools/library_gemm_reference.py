"""How fast does the vendor library run the headline's fp32 GEMM shapes?  (torch.matmul -> hipBLASLt / rocBLAS, fp32 in, fp32 out, no
epilogue.)  A yardstick for the shape limit DESIGN section 7 claims, not a code path of this repository.  GPU box only."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
shapes = [(25216, 1152, 384), (25216, 384, 384), (25216, 1536, 384), (25216, 384, 1536), (12672, 1152, 384), (12672, 384, 384), (12672, 1536, 384),
          (12672, 384, 1536), (36928, 768, 3072), (36928, 3072, 768)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps
print(f"{'M':>6s} {'N':>5s} {'K':>5s}   library us  TF/s    this repo us  TF/s   (y = x W^T, fp32)")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    wt = w.t()
    t_lib = timeit(lambda: torch.matmul(x, wt, out=out))
    t_our = timeit(lambda: ops.linear_fwd(x, w, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:5d} {K:5d}   {t_lib:9.1f} {fl / t_lib / 1e6:6.1f}    {t_our:9.1f} {fl / t_our / 1e6:6.1f}")
