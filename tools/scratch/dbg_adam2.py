import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "dense2sparse-vit_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from tests import cases
from tests.test_model_gpu import build_models, make_args, _t
from oracle import d2s_oracle as O
from d2s.engine import TrainStep
dev = torch.device("cuda:0")
case = cases.MODEL_CASES["micro2"]; cfg = case["cfg"]
student, teacher, sd_s, sd_t = build_models(case, dev)
x, y = _t(cases.make_images(case)), _t(cases.make_labels(case))
hp = dict(lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=1)
ts = TrainStep(student, teacher, make_args(cfg), **hp)
st = O.TrainState({k: _t(v) for k, v in sd_s.items()}, {k: _t(v) for k, v in sd_t.items()}, cfg, **hp)
ts.set_epoch(0); st.set_epoch(0)
for step in (1, 2):
    # gradient of the oracle evaluated AT OUR current parameters (isolates the gradient computation from the optimiser history)
    osd = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in student.named_parameters()}
    tot, _ = O.train_step_losses(osd, {k: _t(v) for k, v in sd_t.items()}, cfg, x, y, warmup=True)
    tot.backward()
    info = ts(x.to(dev), y.to(dev)); oinfo = st.step(x, y)
    rows = []
    for n, p in student.named_parameters():
        if "predictor" not in n: continue
        g_at_ours = osd[n].grad
        if g_at_ours is None or float(g_at_ours.norm()) < 1e-6: continue
        ours = ts.arena.grad_views[n].cpu()
        e1 = float((ours - g_at_ours).norm() / g_at_ours.norm())
        e2 = float((ours - st.sd_s[n].grad).norm() / st.sd_s[n].grad.norm())
        pd = (p.detach().cpu() - st.sd_s[n].detach())
        rows.append((n, e1, e2, float(pd.abs().max()) / hp["lr"], float((pd.abs() > 0.02 * hp["lr"]).float().mean())))
    print(f"--- step {step}: name | grad relerr vs oracle-at-OUR-params | vs oracle trajectory | param maxdiff (lr units) after the step | frac > 0.02 lr")
    for r in rows: print("  %-44s %.1e  %.1e  %.3f  %.4f" % r)
