"""Diagnostic: per-parameter relative gradient error of the HIP path vs an fp64 oracle run (GPU box only)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import numpy as np, torch
from tests import cases
from tests.test_model_gpu import build_models, make_args, _t
from oracle import d2s_oracle as O
from d2s.engine import TrainStep

name = sys.argv[1] if len(sys.argv) > 1 else "small_3stage"
dev = torch.device("cuda:0")
case = cases.MODEL_CASES[name]; cfg = case["cfg"]
student, teacher, sd_s, sd_t = build_models(case, dev)
x, y = _t(cases.make_images(case)), _t(cases.make_labels(case))
ts = TrainStep(student, teacher, make_args(cfg))
student.train()
loss, info = ts.forward_losses(x.to(dev), y.to(dev))
ts.opt.zero_grad(); loss.backward(); torch.cuda.synchronize()
sd64 = {k: _t(v).double().requires_grad_(True) for k, v in sd_s.items()}
t64, i64 = O.train_step_losses(sd64, {k: _t(v).double() for k, v in sd_t.items()}, cfg, x.double(), y)
t64.backward()
print("loss hip %.8f f64 %.8f" % (float(loss), float(t64)))
print("logits err", float((info["logits_s"].detach().cpu().double() - i64["logits_s"]).norm() / i64["logits_s"].norm()))
print("token_s err", float((info["token_s"].detach().cpu().double() - i64["token_s"]).norm() / i64["token_s"].norm()))
for i, pl in enumerate(info["pred_logits"]):
    print("pred_logits", i, float((pl.detach().cpu().double() - i64["pred_logits"][i]).norm() / i64["pred_logits"][i].norm()))
rows = []
for n, p in student.named_parameters():
    g64 = sd64[n].grad.flatten(); d = float(g64.norm()) + 1e-30
    rows.append((float((p.grad.flatten().cpu().double() - g64).norm()) / d, d, n))
for e, d, n in rows:
    if e > 2e-5:
        print("%.2e  norm %.3e  %s" % (e, d, n))

# ---- ReLU-gate hypothesis: is a predictor-0 pre-activation within fp32 noise of zero? ----
import torch.nn.functional as F
cap = {}
orig = student.score_predictor[0].forward_tokens
def rec(xx):
    cap["x"] = xx.detach().cpu().double()
    return orig(xx)
student.score_predictor[0].forward_tokens = rec
with torch.no_grad():
    student(x.to(dev))
sd = {k: _t(v).double() for k, v in sd_s.items()}
p = "score_predictor.0."
h = F.layer_norm(cap["x"][:, 1:], (cfg["dim"],), sd[p + "in_conv.0.weight"], sd[p + "in_conv.0.bias"], 1e-5)
z = F.linear(h, sd[p + "in_conv.1.weight"], sd[p + "in_conv.1.bias"])
print("in_conv   min|z| %.3e  count(|z|<1e-6) %d of %d" % (float(z.abs().min()), int((z.abs() < 1e-6).sum()), z.numel()))
h = F.relu(z); B_, N_, C_ = h.shape
h = torch.cat([h[:, :, :C_ // 2], h[:, :, C_ // 2:].mean(dim=1, keepdim=True).expand(B_, N_, C_ // 2)], dim=-1)
for j, (li, fi) in enumerate([(0, 1), (3, 4), (6, 7), (9, 10), (12, 13)]):
    w = sd[p + f"out_conv.{li}.weight"]
    h = F.layer_norm(h, (w.shape[0],), w, sd[p + f"out_conv.{li}.bias"], 1e-5)
    z = F.linear(h, sd[p + f"out_conv.{fi}.weight"], sd[p + f"out_conv.{fi}.bias"])
    print("layer %d   min|z| %.3e  count(|z|<1e-6) %d of %d" % (j, float(z.abs().min()), int((z.abs() < 1e-6).sum()), z.numel()))
    h = F.relu(z)
