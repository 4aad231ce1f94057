"""GPU tier, dynamic keep ratio (--patch-score-threshold): threshold selection, the fused policy attention
(forward / backward), the training forward and step with the policy in every block, the ragged packed inference, and the mask helpers.

What is pinned by the reference itself: threshold selection (tests/golden/threshold_selection.npz: the reference's four inline calls)
and the whole training-mode forward + the gradients of a linear probe (tests/golden/threshold_<case>.npz: the reference's class run
with patch_score_threshold set).  PARITY UNPINNED (the reference cannot run them, dynamic_vit.py:936, losses.py:81,216-218): the losses
of this mode and the ragged inference - those are checked against the oracle's statement of the build's fix.

Tolerances: masks / counts / packed row order bit-exact; floating point as in test_model_gpu.py (rtol 1e-4); the fused policy backward
omits the O(eps = 1e-6) gradient through the row maximum, far below those tolerances.
"""
import types

import numpy as np
import pytest
import torch

from tests import cases
from oracle import d2s_oracle as O

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _dev():
    return torch.device("cuda:0")


def build_threshold_models(case, device, threshold=True):
    import vit_models
    cfg = case["cfg"]
    common = dict(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                  num_heads=cfg["heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True, num_classes=cfg["num_classes"])
    student = vit_models.VisionTransformerDiffPruning(pruning_loc=list(cfg["pruning_loc"]), token_ratio=list(cfg["token_ratio"]),
                                                      distill=True, topk_selection=True, predictor_loss_type=cfg["loss_type"],
                                                      patch_score_threshold=case["threshold"] if threshold else None, **common)
    teacher = vit_models.VisionTransformerTeacher(**common)
    sd_s, sd_t = cases.make_weights(case)
    student.load_state_dict({k: _t(v) for k, v in sd_s.items()}, strict=True)
    teacher.load_state_dict({k: _t(v) for k, v in sd_t.items()}, strict=True)
    return student.to(device), teacher.to(device), sd_s, sd_t


def test_select_threshold_matches_reference_fixture():
    """Bit-exact against the stable statement of the rule (equal probabilities lowest index first) on every row, and against the
    reference's own output: identical masks wherever the reference's (unstable) torch.sort ordered the ties that straddle the
    threshold the same way, identical kept COUNT and identical multiset of kept probabilities on every row - which is all the
    reference defines for such ties (rows 0-9 of the fixture carry deliberate tie pairs, a constant row and a two-level row)."""
    from d2s import ops
    g = cases.load_golden("threshold_selection")
    exact_rows = tie_rows = 0
    for N in (196, 576, 16):
        pc = _t(cases.make_selection_probs(N))
        p = pc.to(_dev())
        for th in (0.1, 0.35, 0.8):
            mask, counts = ops.select_threshold(p, th)
            got = mask.cpu().numpy() > 0
            stable, scounts = O.select_threshold_stable(pc, th)
            np.testing.assert_array_equal(got, stable.numpy() > 0, err_msg=f"N={N} th={th} (stable rule)")
            assert counts.cpu().tolist() == scounts.tolist()
            want = g[f"mask_{N}_{th}"]
            assert counts.cpu().tolist() == want.sum(axis=1).tolist(), f"N={N} th={th}: kept counts differ from the reference"
            for r in range(want.shape[0]):
                if (got[r] == want[r]).all():
                    exact_rows += 1
                    continue
                tie_rows += 1
                pr = pc[r].numpy()
                np.testing.assert_array_equal(np.sort(pr[got[r]]), np.sort(pr[want[r]]), err_msg=f"N={N} th={th} row {r}: kept values differ")
                assert r < 10 or len(np.unique(pr)) < N, f"N={N} th={th} row {r} differs from the reference without a tie in it"
            pol, _ = ops.select_threshold(p, th, lead=1)          # policy-row form: [1, mask]
            assert pol.shape == (p.shape[0], N + 1) and bool((pol[:, 0] == 1).all())
            np.testing.assert_array_equal(pol[:, 1:].cpu().numpy() > 0, got)
    print(f"[threshold selection] rows bit-identical to the reference: {exact_rows}; rows where only the order of exactly tied tokens differs: {tie_rows}")


@pytest.mark.parametrize("B,n,H", [(2, 17, 2), (3, 197, 6), (2, 99, 3), (1, 577, 2), (2, 33, 1)])
def test_policy_attention_forward_backward(B, n, H):
    """d2s_attn_policy_fwd/bwd vs the oracle's materialised softmax_with_policy (dynamic_vit.py:195-236) and its autograd."""
    from d2s import ops, synth
    D = H * 64
    qkv = _t(synth.normal(f"pa/qkv/{B}/{n}/{H}", (B, n, 3 * D), std=0.7, seed=3))
    pol = _t((synth.normal(f"pa/pol/{B}/{n}", (B, n), seed=4) > 0.3).astype(np.float32))
    pol[:, 0] = 1.0
    if B > 1:
        pol[1, 1:] = 0.0            # an image that keeps only its CLS token: every other row attends to {CLS, itself}
    go = _t(synth.normal(f"pa/go/{B}/{n}/{H}", (B, n, D), std=1.0, seed=5))
    # oracle
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.reshape(B, n, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = O.softmax_with_policy((q @ k.transpose(-2, -1)) * 0.125, pol.unsqueeze(-1))
    ref = (a @ v).transpose(1, 2).reshape(B, n, D)
    ref.backward(go)
    d = _dev()
    qd = qkv.reshape(B * n, 3 * D).to(d)
    out, lse, cinv, cls_row = ops.attn_policy_fwd(qd, pol.to(d), B, n, H, 0.125, want_cls=True)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, n, D), ref.detach().numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(cls_row.cpu().numpy(), a[:, :, 0, :].detach().numpy(), rtol=1e-4, atol=1e-8)
    dq = ops.attn_policy_bwd(qd, pol.to(d), out, go.reshape(B * n, D).to(d), lse, cinv, B, n, H, 0.125)
    got, want = dq.cpu().numpy().reshape(B, n, 3 * D), qr.grad.numpy()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * float(np.abs(want).max()))
    # all-ones policy == the reference's blocks before the first pruning stage: still the eps form, not the plain softmax
    ones = torch.ones(B, n)
    out1, _, _, _ = ops.attn_policy_fwd(qd, ones.to(d), B, n, H, 0.125)
    a1 = O.softmax_with_policy((q @ k.transpose(-2, -1)).detach() * 0.125, ones.unsqueeze(-1))
    np.testing.assert_allclose(out1.cpu().numpy().reshape(B, n, D), (a1 @ v.detach()).transpose(1, 2).reshape(B, n, D).numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("name", list(cases.THRESHOLD_CASES))
def test_threshold_training_forward_matches_reference(name):
    """Student.forward (training, patch_score_threshold set) on the HIP path vs the fixture the reference's own class produced."""
    from d2s import synth
    case = cases.THRESHOLD_CASES[name]
    g = cases.load_golden("threshold_" + name)
    student, _, sd_s, _ = build_threshold_models(case, _dev())
    student.train()
    x = _t(cases.make_images(case)).to(_dev())
    logits, features, pred_logits, masks = student(x)
    assert len(masks) == len(case["cfg"]["pruning_loc"]) == len(pred_logits)
    np.testing.assert_array_equal(masks[-1].cpu().numpy(), g["keep_mask_last"])
    np.testing.assert_allclose(student.keep_ratios.cpu().numpy(), g["keep_ratios"], rtol=1e-6)
    assert abs(student.avg_keep_ratio - float(g["keep_ratios"].mean())) < 1e-6
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    assert list(features.shape) == g["features_shape"].tolist()
    np.testing.assert_allclose(features[:, :4, :16].detach().cpu().numpy(), g["features_slice"], rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(features.detach().double().sum(dim=(1, 2)).cpu().numpy(), g["features_sum"], rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(pred_logits[-1].detach().cpu().numpy(), g["pred_logits_last"], rtol=1e-4, atol=2e-5)
    d = _dev()
    g1 = _t(synth.normal(f"thr/{name}/g1", tuple(logits.shape), seed=case["seed"])).to(d)
    g2 = _t(synth.normal(f"thr/{name}/g2", tuple(features.shape), seed=case["seed"])).to(d)
    g3 = _t(synth.normal(f"thr/{name}/g3", tuple(pred_logits[-1].shape), seed=case["seed"])).to(d)
    torch.autograd.backward([logits, features, pred_logits[-1]], [g1, g2 / features.shape[1], g3])
    params = dict(student.named_parameters())
    # (a) against the reference's fixture: L2 norm and leading elements; (b) the full tensors against an fp64 run of the oracle: the HIP
    # fp32 gradient must be as close to the exact one as the CPU fp32 oracle is (within 4x, floor 2e-4) - both are fp32 evaluations
    # in different summation orders, through up to 12 layers of policy attention
    margins = []

    def probe(sd, dtype):
        lo, fe, pl, _ = O.student_forward_threshold_train(sd, _t(cases.make_images(case)).to(dtype), case["cfg"], case["threshold"], margins)
        ((lo * g1.cpu().to(dtype)).sum() + (fe * g2.cpu().to(dtype)).sum() / fe.shape[1] + (pl[-1] * g3.cpu().to(dtype)).sum()).backward()
    sd32 = {k: _t(v).requires_grad_(True) for k, v in sd_s.items()}
    sd64 = {k: _t(v).double().requires_grad_(True) for k, v in sd_s.items()}
    probe(sd32, torch.float32)
    probe(sd64, torch.float64)
    # a predictor ReLU whose pre-activation is within fp32 rounding of zero is gated by rounding noise; every gradient below that layer
    # then moves by ~1e-3 on either implementation (tests/test_model_gpu.py::test_train_step_parity): wider floor for such cases only
    gate_noise = min(margins) < 5e-6
    floor = 5e-3 if gate_noise else 2e-4
    worst = 0.0
    for n, ref_norm, ref_head in zip([str(s) for s in g["grad_names"]], g["grad_norms"], g["grad_heads"]):
        p = params[n]
        if ref_norm < 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        gf = p.grad.detach().flatten().cpu()
        np.testing.assert_allclose(float(gf.double().norm()), ref_norm, rtol=5e-3 if gate_noise else 2e-3, atol=1e-6, err_msg=n)
        m = min(8, gf.numel())
        np.testing.assert_allclose(gf[:m].numpy(), ref_head[:m], rtol=5e-3, atol=(2e-2 if gate_noise else 5e-3) * float(np.abs(ref_head[:m]).max()) + 2e-6, err_msg=n)
        g64 = sd64[n].grad.flatten()
        denom = float(g64.norm())
        if denom > 1e-6:
            err_hip = float((gf.double() - g64).norm()) / denom
            err_cpu = float((sd32[n].grad.flatten().double() - g64).norm()) / denom
            assert err_hip <= max(4.0 * err_cpu, floor), (n, err_hip, err_cpu, min(margins))
            worst = max(worst, err_hip)
    print(f"[{name}] worst relative gradient error vs the fp64 oracle: {worst:.2e} (relu gate at noise level: {gate_noise}, min margin {min(margins):.1e})")


@pytest.mark.parametrize("name", ["micro_thr1", "micro_thr2"])
def test_threshold_train_step_matches_oracle(name):
    """Full step in threshold mode (teacher fwd, student fwd with the policy in every block, the fixed losses, backward) vs the
    oracle's statement of the same fix.  Parity unpinned for the two losses (see module docstring)."""
    from d2s.engine import TrainStep
    case = cases.THRESHOLD_CASES[name]
    cfg, thr = case["cfg"], case["threshold"]
    student, teacher, sd_s, sd_t = build_threshold_models(case, _dev())
    args = types.SimpleNamespace(keep_ratios=list(cfg["token_ratio"]), mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=thr, step=0)
    ts = TrainStep(student, teacher, args, warmup_steps=0)
    x, y = _t(cases.make_images(case)), _t(cases.make_labels(case))
    student.train()
    loss, info = ts.forward_losses(x.to(_dev()), y.to(_dev()))
    ts.opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    osd = {k: _t(v).requires_grad_(True) for k, v in sd_s.items()}
    tcfg = dict(cfg)
    tcfg["pruning_loc"] = ()
    with torch.no_grad():
        lt, tt, ca = O.teacher_forward({k: _t(v) for k, v in sd_t.items()}, x.clone(), tcfg)
    ls, fs, pl, masks = O.student_forward_threshold_train(osd, x.clone(), cfg, thr)
    ml, accs = O.mask_loss_threshold(pl, ca, masks, thr)
    bl, _, _, tok_kl = O.backbone_loss_threshold(ls, fs, lt, tt, masks, y)
    (ml + bl).backward()
    for a, b in zip(info["kept"], masks):
        np.testing.assert_array_equal(a.cpu().numpy(), b.numpy())
    np.testing.assert_allclose(float(info["mask_loss"]), float(ml), rtol=2e-5)
    np.testing.assert_allclose(float(info["backbone_loss"]), float(bl), rtol=2e-5)
    np.testing.assert_allclose(float(ts.backbone_loss_fn.last_terms[2]), float(tok_kl), rtol=5e-5)
    for i, a in enumerate(accs):
        np.testing.assert_allclose(float(ts.metrics[f"train_mask_acc_{i}"]), float(a), atol=1e-6)
    for n, p in student.named_parameters():
        og = osd[n].grad
        if og is None:
            continue
        assert p.grad is not None, n
        gf, of = p.grad.detach().flatten().cpu().double(), og.flatten().double()
        denom = float(of.norm())
        if denom > 1e-6:
            assert float((gf - of).norm()) / denom < 3e-3, (n, float((gf - of).norm()) / denom)


def test_ragged_pack_and_varlen_attention():
    """d2s_ragged_offsets / d2s_ragged_pack (bit-exact row order) and d2s_attn_varlen_fwd vs per-image dense attention."""
    from d2s import ops, synth
    B, n, H = 5, 50, 2
    D = H * 64
    d = _dev()
    x = _t(synth.normal("rg/x", (B, n, D), seed=1))
    mask = _t((synth.normal("rg/m", (B, n - 1), seed=2) > 0.2).astype(np.float32))
    mask[1] = 0.0                  # an image that keeps nothing but its CLS token
    mask[2] = 1.0                  # an image that keeps everything
    counts = mask.sum(dim=1).int()
    cu = ops.ragged_offsets(counts.to(d), extra=1)
    want_cu = np.concatenate([[0], np.cumsum(counts.numpy() + 1)])
    np.testing.assert_array_equal(cu.cpu().numpy(), want_cu)
    total = int(want_cu[-1])
    packed, src = ops.ragged_pack(x.to(d), mask.to(d), cu, total)
    for b in range(B):
        keep = np.concatenate([[True], mask[b].numpy() > 0])
        np.testing.assert_array_equal(packed[want_cu[b]:want_cu[b + 1]].cpu().numpy(), x[b].numpy()[keep])
        np.testing.assert_array_equal(src[want_cu[b]:want_cu[b + 1]].cpu().numpy(), np.nonzero(keep)[0])
    qkv = _t(synth.normal("rg/qkv", (total, 3 * D), std=0.7, seed=3))
    out, cls_rows = ops.attn_varlen_fwd(qkv.to(d), cu, B, total, n, H, 0.125, want_cls=True)
    for b in range(B):
        seg = qkv[want_cu[b]:want_cu[b + 1]]
        nb = seg.shape[0]
        q, k, v = seg.reshape(1, nb, 3, H, 64).permute(2, 0, 3, 1, 4)
        a = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
        ref = (a @ v).transpose(1, 2).reshape(nb, D)
        np.testing.assert_allclose(out[want_cu[b]:want_cu[b + 1]].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(cls_rows[:, want_cu[b]:want_cu[b + 1]].cpu().numpy(), a[0, :, 0, :].numpy(), rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("name", ["micro_thr1", "small_thr"])
def test_ragged_inference_matches_oracle(name):
    """Eval-mode forward with a dynamic keep ratio: every image continues with its own number of tokens (ragged packed batch) and must
    give what the same blocks give on that image's kept subset alone.  Parity unpinned (the reference raises NameError here)."""
    case = cases.THRESHOLD_CASES[name]
    cfg, thr = case["cfg"], case["threshold"]
    student, _, sd_s, _ = build_threshold_models(case, _dev())
    student.eval()
    x = _t(cases.make_images(case))
    with torch.no_grad():
        logits, cls_attns, pred_logits, masks = student(x.to(_dev()))
    ologits, ofeats, oscores, omask = O.student_forward_threshold_eval({k: _t(v) for k, v in sd_s.items()}, x, cfg, thr)
    np.testing.assert_array_equal(masks[0].cpu().numpy(), omask.numpy())
    counts = omask.sum(dim=1).int().numpy() + 1
    cu = student.cu_seqlens.cpu().numpy()
    np.testing.assert_array_equal(cu, np.concatenate([[0], np.cumsum(counts)]))
    assert len(set(counts.tolist())) > 1 or name == "micro_thr1", "the case should be genuinely ragged"
    np.testing.assert_allclose(logits.cpu().numpy(), ologits.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(pred_logits[0].cpu().numpy(), oscores.numpy(), rtol=1e-4, atol=2e-5)
    for b, f in enumerate(ofeats):
        np.testing.assert_allclose(student.ragged_features[cu[b]:cu[b + 1]].cpu().numpy(), f.numpy(), rtol=1e-4, atol=3e-5)
    loc = cfg["pruning_loc"][0]
    assert len(cls_attns) == cfg["depth"] - 1            # the pruning block itself returns no CLS row, as in the reference (:949)
    assert tuple(cls_attns[0].shape) == (x.shape[0], cfg["heads"], cfg["n_patches"])
    if cfg["depth"] > loc + 1:
        assert tuple(cls_attns[-1].shape) == (cfg["heads"], int(cu[-1]))
        for b in range(x.shape[0]):      # each image's packed CLS row is a probability row
            np.testing.assert_allclose(cls_attns[-1][:, cu[b]:cu[b + 1]].sum(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    # same weights without a threshold but keeping everything == dense path on all tokens: threshold 0 keeps every token
    student.patch_score_threshold = 0.0
    with torch.no_grad():
        l0, _, _, m0 = student(x.to(_dev()))
    assert bool((m0[0] == 1).all())
    ol0, _, _, _ = O.student_forward_threshold_eval({k: _t(v) for k, v in sd_s.items()}, x, cfg, 0.0)
    np.testing.assert_allclose(l0.cpu().numpy(), ol0.numpy(), rtol=1e-4, atol=2e-5)


def test_patch_keep_mask_and_compose_ids():
    """visualizations.py:18-26 (kept / dropped ids -> 0/1 mask) and the composition of stage-relative ids."""
    from d2s import ops
    g = cases.load_golden("selection")
    kept, dropped = _t(g["kept_196_137"]), _t(g["dropped_196_137"])
    want = O.patch_drop_mask(kept, dropped)
    got = ops.patch_keep_mask(kept.to(_dev()), 196)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    rel = _t(g["kept_196_98"])[:, :58].clone() % 137
    rel = torch.sort(rel, dim=1)[0]
    absid = ops.compose_ids(kept.to(_dev()), rel.to(_dev()))
    np.testing.assert_array_equal(absid.cpu().numpy(), torch.gather(kept, 1, rel).numpy())
    import vit_models
    m = vit_models.patch_drop_mask([kept.to(_dev()), rel.to(_dev())], 196)
    assert len(m) == 2 and m[0].shape == (kept.shape[0], 196) and m[1].shape == (kept.shape[0], 196)
    np.testing.assert_array_equal(m[0].cpu().numpy(), want.numpy())
    ref1 = torch.zeros_like(want)
    ref1.scatter_(1, torch.gather(kept, 1, rel), 1)
    np.testing.assert_array_equal(m[1].cpu().numpy(), ref1.numpy())
