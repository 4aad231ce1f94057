import os, sys
sys.path.insert(0, "/root/repo/dense2sparse-vit_amd")
import torch, vit_models
from d2s import ops
dev = torch.device("cuda:0")
B = 128
x = torch.randn(B, 3, 224, 224, device=dev)
ops.set_gemm_mode(ops.GEMM_BF16)
for rep in range(2):
    for name, build in (("keep 0.7", lambda: vit_models.dynamic_vit_small_patch16_224_student([3], [0.7], topk_selection=True, predictor_loss_type="kl_div")),
                        ("keep 0.5", lambda: vit_models.dynamic_vit_small_patch16_224_student([3], [0.5], topk_selection=True, predictor_loss_type="kl_div")),
                        ("keep 0.4", lambda: vit_models.dynamic_vit_small_patch16_224_student([3], [0.4], topk_selection=True, predictor_loss_type="kl_div"))):
        torch.manual_seed(0)
        m = build().to(dev).eval()
        with torch.no_grad():
            for _ in range(3): m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): m(x)
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(f"rep {rep} {name}: {ms:.2f} ms/batch {B/ms*1e3:.0f} img/s", flush=True)
