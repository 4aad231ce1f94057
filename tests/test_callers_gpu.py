"""GPU tier: the caller-level counterparts (train.py::train_one_epoch, evaluate.py::evaluate_performance, utils.py param
groups) run end to end on the accelerated modules, both with the fused TrainStep and with a plain torch.optim.AdamW built the
way the reference builds it (mask_predictor.py:213-230)."""
import os
import types

import numpy as np
import pytest
import torch

from tests import cases
from oracle import d2s_oracle as O
from tests.test_model_gpu import build_models, make_args

pytestmark = pytest.mark.gpu


def _args(cfg, dev):
    a = make_args(cfg)
    a.device, a.warmup_steps, a.weight_decay, a.lr, a.min_lr, a.epochs, a.is_sbatch = dev, 0, 0.05, 5e-4, 1e-5, 25, False
    return a


def test_train_one_epoch_fused_and_torch_optimizer_agree():
    from d2s.engine import TrainStep
    from train import train_one_epoch
    import utils
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro1"]
    cfg = case["cfg"]
    loader = lambda: utils.SyntheticLoader(3, 4, img_size=cfg["img_size"], num_classes=cfg["num_classes"], seed=5, device=dev)
    s1, t1, _, _ = build_models(case, dev)
    a1 = _args(cfg, dev)
    step = TrainStep(s1, t1, a1, lr=a1.lr, min_lr=a1.min_lr, weight_decay=a1.weight_decay, epochs=a1.epochs, warmup_steps=0)
    m1 = train_one_epoch(a1, s1, t1, loader(), step)
    s2, t2, _, _ = build_models(case, dev)
    a2 = _args(cfg, dev)
    groups = utils.get_param_groups(s2, a2)
    opt = torch.optim.AdamW([g for g in groups if g["params"]], lr=a2.lr, weight_decay=a2.weight_decay)
    utils.adjust_learning_rate(opt.param_groups, a2, 0, s2)
    for p in t2.parameters():
        p.requires_grad_(False)
    m2 = train_one_epoch(a2, s2, t2, loader(), opt)
    assert set(m1) == set(m2) and "train_loss" in m1 and "train_mask_loss" in m1 and "train_backbone_loss" in m1
    np.testing.assert_allclose(m1["train_loss"], m2["train_loss"], rtol=1e-4)
    for (n, p), (_, q) in zip(s1.named_parameters(), s2.named_parameters()):
        d = (p.detach() - q.detach()).abs().max().item()
        assert d <= 2 * 3 * a1.lr * 1.01, (n, d)          # Adam: noise-level gradients may move a few elements by lr per step


def test_evaluate_performance_runs_and_reports():
    from evaluate import evaluate_performance
    import utils
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro2"]
    cfg = case["cfg"]
    s, t, _, _ = build_models(case, dev)
    a = _args(cfg, dev)
    m = evaluate_performance(a, s, t, utils.SyntheticLoader(2, 4, img_size=cfg["img_size"], num_classes=cfg["num_classes"], seed=3, device=dev))
    for k in ("val_loss", "val_acc", "unpruned_acc", "val_mask_loss", "val_mask_acc_0", "val_mask_acc_1"):
        assert k in m, k
    assert 0.0 <= m["val_acc"] <= 1.0 and m["val_acc"] == m["unpruned_acc"] and np.isfinite(m["val_loss"])


def test_mask_predictor_cli_runs_the_epoch_loop(capsys):
    """The entry script's sequence (factories -> groups -> per epoch LR/freeze schedule -> train -> evaluate) on DeiT-Ti with the
    reference's flags; epoch 0 is a warm-up epoch (backbone frozen), epoch 1 trains everything."""
    import mask_predictor
    best = mask_predictor.main(["--arch", "deit_tiny", "--pruning-locs", "3", "--keep-ratios", "0.5", "--epochs", "2", "--warmup-steps", "1",
                                "--batch-size", "4", "--steps-per-epoch", "2", "--val-steps", "1", "--topk-selection"])
    out = capsys.readouterr().out
    assert 0.0 <= best <= 1.0
    assert "Epoch 2/2" in out and "Training complete" in out and "train images/s" in out


def test_mask_predictor_cli_with_dynamic_keep_ratio(capsys):
    """--patch-score-threshold through the entry script: policy-masked training steps (incl. a warm-up epoch), then the ragged
    inference of evaluate_performance."""
    import mask_predictor
    best = mask_predictor.main(["--arch", "deit_tiny", "--pruning-locs", "3", "--keep-ratios", "0.5", "--epochs", "2", "--warmup-steps", "1",
                                "--batch-size", "4", "--steps-per-epoch", "2", "--val-steps", "1", "--topk-selection", "--patch-score-threshold", "0.3"])
    out = capsys.readouterr().out
    assert 0.0 <= best <= 1.0
    assert "Epoch 2/2" in out and "Training complete" in out


def test_epoch_callers_with_a_shorter_last_batch_in_threshold_mode():
    """train_one_epoch / evaluate_performance over loaders whose last batch is shorter (drop_last=False, ddp_training.py:15-20) with
    --patch-score-threshold set: the per-image keep ratios of all batches end up in the min / avg / max metrics."""
    import vit_models
    import utils
    from d2s.engine import TrainStep
    from train import train_one_epoch
    from evaluate import evaluate_performance
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = dict(img_size=64, embed_dim=128, depth=3, num_heads=2, num_classes=10)
    s = vit_models.VisionTransformerDiffPruning(pruning_loc=[1], token_ratio=[0.5], distill=True, topk_selection=True,
                                                predictor_loss_type="kl_div", patch_score_threshold=0.3, **kw).to(dev)
    t = vit_models.VisionTransformerTeacher(**kw).to(dev)
    args = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=0.3, step=0, warmup_steps=0,
                                 device=dev, is_sbatch=False)
    step = TrainStep(s, t, args)
    m = train_one_epoch(args, s, t, utils.SyntheticLoader(3, 4, 64, 10, seed=1, device=dev, last_batch=3), step, None)
    m.update(evaluate_performance(args, s, t, utils.SyntheticLoader(3, 4, 64, 10, seed=2, device=dev, last_batch=1)))
    for pre in ("train", "val"):
        lo, av, hi = m[f"{pre}_min_keep_ratio"], m[f"{pre}_avg_keep_ratio"], m[f"{pre}_max_keep_ratio"]
        assert 0.0 < lo <= av <= hi <= 1.0, (pre, lo, av, hi)


def test_mask_loss_mse_branch_matches_reference_fixture():
    """losses.MaskLoss(mask_loss_type='mse') on the HIP path (d2s_kl_rows mode 3 + d2s_gather_renorm) against the reference's own
    output: loss, gradients of both stages' scores, and the metrics keys it writes."""
    from losses import MaskLoss
    dev = torch.device("cuda:0")
    g = cases.load_golden("mask_loss_mse")
    p0 = torch.from_numpy(g["p0"]).to(dev).requires_grad_(True)
    p1 = torch.from_numpy(g["p1"]).to(dev).requires_grad_(True)
    args = types.SimpleNamespace(keep_ratios=[0.7, 0.35], mask_loss_type="mse")
    metrics = {}
    loss = MaskLoss(args, "train")([p0, p1], torch.from_numpy(g["cls_attn"]).to(dev),
                                   [torch.from_numpy(g["kept0"]).to(dev), torch.from_numpy(g["kept1"]).to(dev)], metrics)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=2e-5)
    np.testing.assert_allclose(p0.grad.cpu().numpy(), g["g0"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(p1.grad.cpu().numpy(), g["g1"], rtol=1e-4, atol=1e-7)
    assert sorted(metrics) == [str(k) for k in g["metric_keys"]]


def test_early_exit_head_is_never_updated():
    """--early-exit: the extra head exists (keys, 'early_exit' group) but nothing calls it; like torch's AdamW on grad-None parameters,
    the fused optimiser must leave it untouched while everything else trains."""
    import vit_models
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = dict(img_size=64, embed_dim=128, depth=3, num_heads=2, num_classes=10)
    s = vit_models.VisionTransformerDiffPruning(pruning_loc=[1], token_ratio=[0.05], distill=True, topk_selection=True,
                                                predictor_loss_type="kl_div", early_exit=True, **kw).to(dev)
    t = vit_models.VisionTransformerTeacher(**kw).to(dev)
    args = types.SimpleNamespace(keep_ratios=[0.05], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    ts = TrainStep(s, t, args, lr=1e-3, weight_decay=0.05, warmup_steps=0)
    before = {k: v.clone() for k, v in s.state_dict().items()}
    x = torch.randn(4, 3, 64, 64, device=dev)
    y = torch.randint(0, 10, (4,), device=dev)
    for _ in range(2):
        ts(x, y)
    after = s.state_dict()
    for k in before:
        if k.startswith("early_exit_head."):
            assert torch.equal(before[k], after[k]), k
    assert not torch.equal(before["blocks.0.mlp.fc1.weight"], after["blocks.0.mlp.fc1.weight"])


def test_soft_target_cross_entropy_under_mixup():
    """BackboneLoss with args.mixup > 0 uses the soft-target cross entropy (losses.py:170-172; timm's SoftTargetCrossEntropy =
    mean_b sum_c -t log_softmax(x)) on [B, classes] label distributions from the caller's mixup_fn.  Parity: the formula itself (timm is
    not in this image, so the reference side of this one term is restated, not imported); loss and gradient vs torch on the CPU."""
    from losses import BackboneLoss
    dev = torch.device("cuda:0")
    B, C, k, D = 6, 10, 5, 32
    g = torch.Generator().manual_seed(3)
    logits_s = torch.randn(B, C, generator=g)
    logits_t = torch.randn(B, C, generator=g)
    tok_s = torch.randn(B, k, D, generator=g)
    tok_t = torch.randn(B, 12, D, generator=g)
    kept = torch.stack([torch.randperm(12, generator=g)[:k].sort().values for _ in range(B)])
    lam = 0.3
    y1, y2 = torch.randint(0, C, (B,), generator=g), torch.randint(0, C, (B,), generator=g)
    soft = lam * torch.nn.functional.one_hot(y1, C).float() + (1 - lam) * torch.nn.functional.one_hot(y2, C).float()
    soft = soft * 0.9 + 0.1 / C                                   # label smoothing as the mixup transform applies it
    ls_ref = logits_s.clone().requires_grad_(True)
    ts_ref = tok_s.clone().requires_grad_(True)
    ref, _, _, _ = O.backbone_loss(ls_ref, ts_ref, logits_t, tok_t, [kept], soft)
    ref.backward()
    args = types.SimpleNamespace(mixup=0.8, patch_score_threshold=None)
    ls = logits_s.to(dev).requires_grad_(True)
    tsd = tok_s.to(dev).requires_grad_(True)
    loss = BackboneLoss(args)(ls, tsd, logits_t.to(dev), tok_t.to(dev), [kept.to(dev)], soft.to(dev), {})
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-5)
    np.testing.assert_allclose(ls.grad.cpu().numpy(), ls_ref.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(tsd.grad.cpu().numpy(), ts_ref.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the driver's contract keys plus the `roofline` / `cpu_baseline`-style objects of this
    scope (a short run: 2 timed steps, no CPU leg), and the per-kernel figures hang together (achieved <= peak, frac = achieved / peak, the
    dominant kernel's launches per step is what the model issues)."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(cases.REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "host_enqueue_ms_per_step", "c_abi_calls_per_step"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "images/s" and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 128 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["achieved"] <= r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches_per_step"] == 159.0
    assert d["gather"]["bound"] == "hbm" and 0 < d["gather"]["frac"] < 1 and d["c_abi_calls_per_step"] < 200
