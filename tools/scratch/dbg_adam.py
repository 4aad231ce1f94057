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
for warm in (1, 0):
    student, teacher, sd_s, sd_t = build_models(case, dev)
    x, y = _t(cases.make_images(case)), _t(cases.make_labels(case))
    hp = dict(lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=warm)
    ts = TrainStep(student, teacher, make_args(cfg), **hp)
    st = O.TrainState({k: _t(v) for k, v in sd_s.items()}, {k: _t(v) for k, v in sd_t.items()}, cfg, **hp)
    k = 0
    for epoch in (0, 1):
        ts.set_epoch(epoch); st.set_epoch(epoch)
        for _ in range(2):
            info = ts(x.to(dev), y.to(dev)); oinfo = st.step(x, y); k += 1
            worst = []
            for n, p in student.named_parameters():
                ref = st.sd_s[n].detach().numpy(); got = p.detach().cpu().numpy()
                bad = ~np.isclose(got, ref, rtol=2e-4, atol=2e-6)
                g_ref = st.sd_s[n].grad
                gd = ts.arena.grad_views[n].cpu().numpy()
                gerr = float(np.linalg.norm(gd - g_ref.numpy()) / (np.linalg.norm(g_ref.numpy()) + 1e-30)) if g_ref is not None else -1
                if g_ref is None or float(g_ref.double().norm()) < 1e-6:
                    continue
                worst.append((float(bad.mean()), float(np.abs(got - ref).max()) / hp["lr"], gerr, n))
            worst.sort(reverse=True)
            print(f"warm={warm} step {k} epoch {epoch} loss {float(info['loss']):.6f} vs {float(oinfo['loss']):.6f}; worst bad-frac / maxdiff(lr units) / grad relerr:", [(round(a, 5), round(b, 3), f"{c:.1e}", n) for a, b, c, n in worst[:4]])
