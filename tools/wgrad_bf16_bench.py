"""bf16-mode weight gradient per-launch time at config 5's shapes (GPU box only): both operands in bf16 (+ the fp32 gradient for the bias).
Run once with D2S_TN_TOKEN_MAJOR=1 (token-major matrix kernel, no transposing pass) and once with 0 (transposing split + pieces kernel)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
from d2s import ops
dev = torch.device("cuda:0")
ops.set_gemm_mode(ops.GEMM_BF16)
print("D2S_TN_TOKEN_MAJOR =", os.environ.get("D2S_TN_TOKEN_MAJOR", "1 (default)"))
for tokens in (36928, 11072):
    for n_out, n_in, both in ((2304, 768, False), (768, 768, True), (3072, 768, False), (768, 3072, True)):
        dy = torch.randn(tokens, n_out, device=dev)
        x16 = torch.randn(tokens, n_in, device=dev).bfloat16()
        dy16 = dy.bfloat16()
        dW, db = torch.empty(n_out, n_in, device=dev), torch.empty(n_out, device=dev)
        def run():
            ops.linear_wgrad(dy if both else None, None, dW, db=db, x16=x16, dy16=dy16)
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 20
        print(f"tokens {tokens:6d}  dW {n_out:4d} x {n_in:4d}  dy {'fp32+bf16' if both else 'bf16 only'}  {us:8.1f} us  {2.0 * tokens * n_out * n_in / us / 1e6:7.1f} TF/s (whole call: passes + matrix kernel + slab combine)")
