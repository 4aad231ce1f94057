"""Inference throughput of the pruned student vs the dense backbone (eval mode, forward only, fp32, B=128, DeiT-S 224) - what the token
pruning buys at deployment (evaluate.py path).  GPU box only."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch
import vit_models
from d2s import ops

dev = torch.device("cuda:0")
B = 128
x = torch.randn(B, 3, 224, 224, device=dev)
rows = []
for mode, mname in ((ops.GEMM_EXACT, "fp32 exact"), (ops.GEMM_SPLIT, "bf16x3 split"), (ops.GEMM_BF16, "bf16 operands")):
    ops.set_gemm_mode(mode)
    for name, build in (("dense teacher (197 tokens throughout)", lambda: vit_models.dynamic_vit_small_patch16_224_teacher()),
                        ("student keep 0.7 @ block 3", lambda: vit_models.dynamic_vit_small_patch16_224_student([3], [0.7], topk_selection=True, predictor_loss_type="kl_div")),
                        ("student keep 0.5 @ block 3", lambda: vit_models.dynamic_vit_small_patch16_224_student([3], [0.5], topk_selection=True, predictor_loss_type="kl_div")),
                        ("student keep 0.7/0.5/0.3 @ blocks 3,6,9", lambda: vit_models.dynamic_vit_small_patch16_224_student([3, 6, 9], [0.7, 0.5, 0.3], topk_selection=True, predictor_loss_type="kl_div"))):
        torch.manual_seed(0)
        m = build().to(dev).eval()
        with torch.no_grad():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                m(x)
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        rows.append((mname, name, ms, B / ms * 1e3))
        print(f"{mname:14s} {name:42s} {ms:7.2f} ms/batch  {B / ms * 1e3:9.0f} images/s", flush=True)
ops.set_gemm_mode(ops.GEMM_EXACT)
