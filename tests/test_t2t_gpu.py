"""GPU tier, T2T path: the HIP T2T modules against the oracle and the reference-generated fixture, plus the
build-defined pruned T2T train step (BASELINE config 4 composition) against the oracle's composition of the same pieces."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import cases
from oracle import d2s_oracle as O

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_unfold_fwd_bwd_both_layouts():
    from d2s import functional_t2t as TF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for (B, C, H, k, s, p) in ((2, 3, 32, 7, 4, 2), (2, 64, 8, 3, 2, 1), (1, 5, 9, 3, 2, 1)):
        x = torch.randn(B, C, H, H, generator=g)
        xr = x.clone().requires_grad_(True)
        ref = F.unfold(xr, kernel_size=k, stride=s, padding=p).transpose(1, 2)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd = x.to(dev).requires_grad_(True)
        out = TF.UnfoldFn.apply(xd, k, s, p, None)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.detach().numpy())
        out.backward(go.to(dev))
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)
        # token layout [B, H*W, C] read in place == transpose/reshape to an image first (t2t_vit.py:90,97)
        tok = x.flatten(2).transpose(1, 2).contiguous()
        td = tok.to(dev).requires_grad_(True)
        out2 = TF.UnfoldFn.apply(td, k, s, p, H)
        np.testing.assert_array_equal(out2.detach().cpu().numpy(), ref.detach().numpy())
        out2.backward(go.to(dev))
        np.testing.assert_allclose(td.grad.cpu().numpy(), xr.grad.flatten(2).transpose(1, 2).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tt", ["performer", "transformer"])
def test_t2t_vit_matches_reference_fixture(tt):
    import vit_models
    dev = torch.device("cuda:0")
    g = cases.load_golden("t2t")
    c = cases.T2T_CASE
    m = vit_models.T2T_ViT(img_size=c["img_size"], tokens_type=tt, embed_dim=c["dim"], depth=c["depth"], num_heads=c["heads"],
                           mlp_ratio=c["mlp_ratio"], num_classes=c["num_classes"], token_dim=64)
    sd = cases.make_t2t_weights(tt)
    assert list(m.state_dict().keys()) == list(sd.keys())           # reference key names and order (checked in gen_golden)
    m.load_state_dict({k: _t(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x = _t(cases.make_t2t_images()).to(dev)
    tok0 = m.tokens_to_token.soft_split0(x)
    np.testing.assert_array_equal(tok0.cpu().numpy(), g[f"{tt}_unfold0"])
    a1 = m.tokens_to_token.attention1(tok0)
    np.testing.assert_allclose(a1.detach().cpu().numpy(), g[f"{tt}_attention1"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(m.tokens_to_token(x).detach().cpu().numpy(), g[f"{tt}_t2t_module"], rtol=1e-4, atol=2e-5)
    cls_feat, heads = m.forward_features(x)
    logits = m(x)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[f"{tt}_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(heads[-1].detach().cpu().numpy(), g[f"{tt}_block_head_last"], rtol=1e-4, atol=3e-5)
    from d2s import synth
    gl = _t(synth.normal("t2t/g", tuple(logits.shape), seed=9)).to(dev)
    m.zero_grad()
    (logits * gl).sum().backward()
    params = dict(m.named_parameters())
    for n, ref in zip([str(s) for s in g[f"{tt}_grad_names"]], g[f"{tt}_grad_norms"]):
        if ref < 0:
            assert params[n].grad is None, n
            continue
        np.testing.assert_allclose(float(params[n].grad.double().norm()), ref, rtol=1e-3, atol=1e-7, err_msg=n)
    # full-tensor check of every gradient against the oracle
    osd = {k: _t(v).requires_grad_(params[k].requires_grad if k in params else False) for k, v in sd.items()}
    ol, _ = O.t2t_forward(osd, x.cpu(), c["depth"], c["heads"], tt)
    (ol * gl.cpu()).sum().backward()
    for n, p in params.items():
        if p.grad is None:
            continue
        og = osd[n].grad
        assert float((p.grad.cpu().double() - og.double()).norm()) <= 2e-4 * float(og.double().norm()) + 1e-7, n


def _oracle_pruned_t2t(sd, x, c, pruning_loc, ratios, training=True):
    """The build-defined composition, restated with oracle pieces: T2T front end + predictor / top-k / gather + plain blocks."""
    cfg = O.make_cfg(dim=c["dim"], pruning_loc=pruning_loc, token_ratio=ratios)
    B = x.shape[0]
    t = O.t2t_module(sd, x, "performer")
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    pred_logits, kept_all, rows = [], [], []
    s = 0
    for i in range(c["depth"]):
        if i in pruning_loc:
            scores, probs = O.predictor(sd, s, t[:, 1:], cfg)
            kept, _ = O.select_topk(probs, int(196 * ratios[s]))
            pred_logits.append(scores)
            kept_all.append(kept)
            t = O.gather_pack(t, kept)
            s += 1
        t, row = O.plain_block(sd, i, t, c["heads"], want_cls=True)
        rows.append(row)
    f = F.layer_norm(t, (c["dim"],), sd["norm.weight"], sd["norm.bias"], 1e-5)
    logits = F.linear(f[:, 0], sd["head.weight"], sd["head.bias"])
    return logits, f[:, 1:], pred_logits, kept_all, rows


def test_pruned_t2t_train_step_matches_oracle_composition():
    import vit_models
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    c = cases.T2T_CASE
    loc, ratios = (1,), (0.05,)
    sd_s = cases.make_t2t_weights("performer", pruning_loc=loc)
    sd_t = cases.make_t2t_weights("performer")
    kw = dict(img_size=c["img_size"], embed_dim=c["dim"], depth=c["depth"], num_heads=c["heads"], mlp_ratio=c["mlp_ratio"],
              num_classes=c["num_classes"], token_dim=64, tokens_type="performer")
    student = vit_models.T2T_ViT_DiffPruning(pruning_loc=list(loc), token_ratio=list(ratios), **kw)
    teacher = vit_models.T2T_ViT_Teacher(**kw)
    student.load_state_dict({k: _t(v) for k, v in sd_s.items()})
    teacher.load_state_dict({k: _t(v) for k, v in sd_t.items()})
    student, teacher = student.to(dev), teacher.to(dev)
    args = types.SimpleNamespace(keep_ratios=list(ratios), mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    ts = TrainStep(student, teacher, args)
    x = _t(cases.make_t2t_images())
    y = torch.tensor([3, 7])
    student.train()
    loss, info = ts.forward_losses(x.to(dev), y.to(dev))
    ts.opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    # oracle
    osd = {k: _t(v).requires_grad_(k not in ("pos_embed",) and not k.endswith(".w")) for k, v in sd_s.items()}
    tsd = {k: _t(v) for k, v in sd_t.items()}
    with torch.no_grad():
        lt, tok_t, _, _, rows_t = _oracle_pruned_t2t(tsd, x, c, (), ())
        cls_attn = torch.stack(rows_t, dim=1)
    ls, tok_s, pl, kept, _ = _oracle_pruned_t2t(osd, x, c, loc, ratios)
    ml, _ = O.mask_loss_kl(pl, cls_attn, kept, list(ratios))
    bl, _, _, _ = O.backbone_loss(ls, tok_s, lt, tok_t, kept, y)
    (ml + bl).backward()
    for a, b in zip(info["kept"], kept):
        np.testing.assert_array_equal(a.cpu().numpy(), b.numpy())
    np.testing.assert_allclose(info["logits_s"].detach().cpu().numpy(), ls.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(info["mask_loss"]), float(ml), rtol=5e-5)
    np.testing.assert_allclose(float(info["backbone_loss"]), float(bl), rtol=5e-5)
    for n, p in student.named_parameters():
        og = osd[n].grad
        if og is None:
            assert p.grad is None or not p.requires_grad, n
            continue
        assert p.grad is not None, n
        assert float((p.grad.cpu().double() - og.double()).norm()) <= 3e-4 * float(og.double().norm()) + 1e-6, n


@pytest.mark.parametrize("tt", ["performer", "transformer"])
def test_t2t_vit_14_at_224_matches_reference_fixture(tt):
    """BASELINE config 4 at its own geometry: T2T-ViT-14 (D 384, depth 14, 6 heads, mlp_ratio 3) on a 224x224 image - the 3136-token
    soft split, the 3136-token performer / transformer stage, the 784-token stage, the 14-block backbone - on the HIP path against
    the fixture the reference's own T2T_ViT produced (tests/golden/t2t_224.npz), incl. every parameter-gradient norm."""
    import vit_models
    from d2s import synth
    dev = torch.device("cuda:0")
    g = cases.load_golden("t2t_224")
    c = cases.T2T_224_CASE
    m = vit_models.T2T_ViT(img_size=c["img_size"], tokens_type=tt, embed_dim=c["dim"], depth=c["depth"], num_heads=c["heads"],
                           mlp_ratio=c["mlp_ratio"], num_classes=c["num_classes"], token_dim=64)
    sd = cases.make_t2t_weights(tt, case=c)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: _t(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x = _t(cases.make_t2t_images(c)).to(dev)
    tok0 = m.tokens_to_token.soft_split0(x)
    assert list(tok0.shape) == g[f"{tt}_unfold0_shape"].tolist()
    np.testing.assert_array_equal(tok0[:, 1000:1004].cpu().numpy(), g[f"{tt}_unfold0_slice"])
    a1 = m.tokens_to_token.attention1(tok0)
    np.testing.assert_allclose(a1[:, ::392].detach().cpu().numpy(), g[f"{tt}_attention1_slice"], rtol=1e-4, atol=2e-5)
    tm = m.tokens_to_token(x)
    assert list(tm.shape) == g[f"{tt}_t2t_module_shape"].tolist()
    np.testing.assert_allclose(tm[:, ::28, ::8].detach().cpu().numpy(), g[f"{tt}_t2t_module_slice"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(tm.detach().double().sum(dim=2).cpu().numpy(), g[f"{tt}_t2t_module_sum"], rtol=1e-4, atol=2e-3)
    cls_feat, heads = m.forward_features(x)
    logits = m(x)
    assert len(heads) == int(g[f"{tt}_n_block_heads"])
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[f"{tt}_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(heads[-1][:, ::16].detach().cpu().numpy(), g[f"{tt}_block_head_last_slice"], rtol=1e-4, atol=3e-5)
    gl = _t(synth.normal("t2t224/g", tuple(logits.shape), seed=9)).to(dev)
    m.zero_grad()
    (logits * gl).sum().backward()
    params = dict(m.named_parameters())
    for n, ref in zip([str(s) for s in g[f"{tt}_grad_names"]], g[f"{tt}_grad_norms"]):
        if ref < 0:
            assert params[n].grad is None, n
            continue
        np.testing.assert_allclose(float(params[n].grad.double().norm()), ref, rtol=2e-3, atol=1e-7, err_msg=n)


def test_pruned_t2t_14_full_size_step_properties():
    """BASELINE config 4 as bench.py --config c4 runs it (pruned T2T-ViT-14, keep 0.5 @ block 3, batch 64 at 224x224), where the oracle
    is too slow to be the checker: the whole train step is bit-identical between two runs, kept / dropped ids are sorted, unique, in
    range and partition the 196 tokens, the kept count is int(196 * 0.5), and the first 32 images give the same logits and ids alone
    or inside the batch of 64 (batch independence: the property data parallelism rests on)."""
    import vit_models
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    student = vit_models.t2t_vit_14_student([3], [0.5]).to(dev)
    teacher = vit_models.t2t_vit_14_teacher().to(dev)
    args = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    ts = TrainStep(student, teacher, args)
    B = 64
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((B, 3, 224, 224), device=dev, generator=g)
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)
    runs = []
    for _ in range(2):
        student.train()
        loss, info = ts.forward_losses(x, y)
        ts.opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        runs.append((float(loss), info["kept"][0].clone(), info["logits_s"].detach().clone(), ts.arena.grads.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert torch.equal(runs[0][3], runs[1][3]), "gradients differ between two identical steps"
    assert np.isfinite(runs[0][0])
    kept, dropped = runs[0][1].cpu().numpy(), student.dropped_token_indices[0].cpu().numpy()
    assert kept.shape == (B, 98) and dropped.shape == (B, 98)
    assert (np.diff(kept, axis=1) > 0).all() and (np.diff(dropped, axis=1) > 0).all()
    both = np.sort(np.concatenate([kept, dropped], axis=1), axis=1)
    np.testing.assert_array_equal(both, np.tile(np.arange(196), (B, 1)))
    with torch.no_grad():
        _, half = ts.forward_losses(x[:32].contiguous(), y[:32].contiguous())
    np.testing.assert_array_equal(half["kept"][0].cpu().numpy(), kept[:32])
    np.testing.assert_allclose(half["logits_s"].cpu().numpy(), runs[0][2][:32].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_pruned_t2t_14_bf16_mode_step():
    """Config 4's model through the bf16 arithmetic mode and its bf16 data path (the backbone blocks are the same BlockFn as DeiT's;
    the T2T stages keep their fp32 kernels): two identical steps are bit-identical, the teacher's logits stay within bf16 distance of the
    exact mode's (rtol 2e-2 of the largest logit), and the kept / dropped ids still partition the 196 tokens."""
    import vit_models
    from d2s import ops
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    args = types.SimpleNamespace(keep_ratios=[0.5], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    B = 16
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn((B, 3, 224, 224), device=dev, generator=g)
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)

    def run(mode, reps):
        ops.set_gemm_mode(mode)
        try:
            torch.manual_seed(0)
            student = vit_models.t2t_vit_14_student([3], [0.5]).to(dev)
            teacher = vit_models.t2t_vit_14_teacher().to(dev)
            ts = TrainStep(student, teacher, args)
            out = []
            for _ in range(reps):
                student.train()
                loss, info = ts.forward_losses(x, y)
                ts.opt.zero_grad()
                loss.backward()
                torch.cuda.synchronize()
                out.append((float(loss.detach()), info["kept"][0].clone(), info["logits_t"].clone(), ts.arena.grads.clone(),
                            student.dropped_token_indices[0].clone()))
        finally:
            ops.set_gemm_mode(ops.GEMM_EXACT)
        return out

    exact = run(ops.GEMM_EXACT, 1)
    bf = run(ops.GEMM_BF16, 2)
    assert bf[0][0] == bf[1][0] and torch.equal(bf[0][1], bf[1][1]) and torch.equal(bf[0][3], bf[1][3]), "bf16 mode is not deterministic"
    assert np.isfinite(bf[0][0]) and torch.isfinite(bf[0][3]).all() and float(bf[0][3].abs().max()) > 0
    lt = exact[0][2].cpu().numpy()
    np.testing.assert_allclose(bf[0][2].cpu().numpy(), lt, rtol=2e-2, atol=2e-2 * float(np.abs(lt).max()))
    np.testing.assert_allclose(bf[0][0], exact[0][0], rtol=2e-2)
    kept, dropped = bf[0][1].cpu().numpy(), bf[0][4].cpu().numpy()
    np.testing.assert_array_equal(np.sort(np.concatenate([kept, dropped], axis=1), axis=1), np.tile(np.arange(196), (B, 1)))
