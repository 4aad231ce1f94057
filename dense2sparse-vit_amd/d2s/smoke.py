"""One small invocation of the hot path on the GPU, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import os
import sys
import types

import numpy as np
import torch


def run(device):
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import d2s_oracle as O          # the checker (test infrastructure)
    from tests import cases
    import vit_models
    from d2s.engine import TrainStep

    case = cases.MODEL_CASES["micro2"]
    cfg = case["cfg"]
    common = dict(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                  num_heads=cfg["heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True, num_classes=cfg["num_classes"])
    student = vit_models.VisionTransformerDiffPruning(pruning_loc=list(cfg["pruning_loc"]), token_ratio=list(cfg["token_ratio"]),
                                                      distill=True, topk_selection=True, predictor_loss_type="kl_div", **common)
    teacher = vit_models.VisionTransformerTeacher(**common)
    sd_s, sd_t = cases.make_weights(case)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    student.load_state_dict({k: t(v) for k, v in sd_s.items()})
    teacher.load_state_dict({k: t(v) for k, v in sd_t.items()})
    student, teacher = student.to(device), teacher.to(device)
    args = types.SimpleNamespace(keep_ratios=list(cfg["token_ratio"]), mask_loss_type="kl_div", mixup=0.0,
                                 patch_score_threshold=None, step=0)
    ts = TrainStep(student, teacher, args)
    x, y = t(cases.make_images(case)), t(cases.make_labels(case))
    info = ts(x.to(device), y.to(device))
    torch.cuda.synchronize()
    osd = {k: t(v).requires_grad_(True) for k, v in sd_s.items()}
    ototal, oinfo = O.train_step_losses(osd, {k: t(v) for k, v in sd_t.items()}, cfg, x, y)
    for a, b in zip(info["kept"], oinfo["kept"]):
        assert np.array_equal(a.cpu().numpy(), b.numpy()), "kept-token ids differ from the oracle"
    np.testing.assert_allclose(float(info["loss"]), float(ototal), rtol=5e-5)
    np.testing.assert_allclose(info["logits_s"].detach().cpu().numpy(), oinfo["logits_s"].detach().numpy(), rtol=1e-4, atol=2e-5)
    print(f"[smoke] ok: loss {float(info['loss']):.6f} (oracle {float(ototal):.6f}); kept ids bit-exact; "
          f"lib {os.path.basename(__import__('d2s.lib', fromlist=['x']).LIB_PATH)}")
