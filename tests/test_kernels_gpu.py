import math
import os
"""GPU tier, per-kernel parity: every HIP entry point (called through the C ABI) against the CPU oracle / plain torch
fp32 CPU ops on the same seeded inputs.  Integer outputs must be bit-exact; floating point within the tolerance
written at each assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import cases
from oracle import d2s_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rand(name, shape, std=1.0, seed=1):
    from d2s import synth
    return torch.from_numpy(synth.normal(name, shape, std=std, seed=seed))


@pytest.fixture(scope="module")
def ops():
    from d2s import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(197 * 2, 1152, 384), (99 * 3, 384, 1536), (128, 1000, 384), (300, 96, 192),
                                   (50, 1, 96), (4, 10, 128), (391, 200, 36), (1024, 1536, 384)])
def test_gemm_nt_bias(ops, M, N, K):
    x, w, b = _rand("gx", (M, K)), _rand("gw", (N, K), 0.05), _rand("gb", (N,), 0.1)
    ref = F.linear(x, w, b)
    got = ops.linear_fwd(x.to(_dev()), w.to(_dev()), b.to(_dev())).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_gemm_epilogues(ops):
    M, N, K = 333, 520, 136
    x, w, b, r = _rand("ex", (M, K)), _rand("ew", (N, K), 0.1), _rand("eb", (N,), 0.1), _rand("er", (M, N))
    xd, wd, bd, rd = (t.to(_dev()) for t in (x, w, b, r))
    z = F.linear(x, w, b)
    np.testing.assert_allclose(ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_RELU).cpu().numpy(), F.relu(z).numpy(), rtol=1e-4, atol=2e-5)
    pre = torch.empty((M, N), device=_dev())
    g = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_GELU, aux_out=pre).cpu()
    np.testing.assert_allclose(g.numpy(), F.gelu(z).numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(pre.cpu().numpy(), z.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_RESID, aux=rd).cpu().numpy(), (z + r).numpy(), rtol=1e-4, atol=2e-5)
    # dgrad with activation-gradient epilogues
    dy = _rand("edy", (M, N))
    w2 = _rand("ew2", (N, K), 0.1)       # dx[M,K] = dy @ w2
    pre_k = _rand("epk", (M, K))
    dx = dy @ w2
    zz = pre_k.clone().requires_grad_(True)
    F.gelu(zz).backward(torch.ones_like(zz))
    got = ops.linear_dgrad(dy.to(_dev()), w2.to(_dev()), epi=ops.EPI_MUL_GELU_GRAD, aux=pre_k.to(_dev())).cpu()
    np.testing.assert_allclose(got.numpy(), (dx * zz.grad).numpy(), rtol=1e-4, atol=3e-5)
    got = ops.linear_dgrad(dy.to(_dev()), w2.to(_dev()), epi=ops.EPI_MUL_RELU_MASK, aux=pre_k.to(_dev())).cpu()
    np.testing.assert_allclose(got.numpy(), (dx * (pre_k > 0)).numpy(), rtol=1e-4, atol=3e-5)
    got = ops.linear_dgrad(dy.to(_dev()), w2.to(_dev())).cpu()
    np.testing.assert_allclose(got.numpy(), dx.numpy(), rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("M,N,K", [(197 * 8, 1536, 384), (99 * 5, 384, 1536), (700, 10, 128), (5000, 1, 96), (64, 96, 192)])
def test_gemm_wgrad_and_colsum(ops, M, N, K):
    dy, x = _rand("wdy", (M, N), 0.5), _rand("wx", (M, K))
    ref = dy.t().double() @ x.double()
    dW = torch.zeros((N, K), device=_dev())
    ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW)
    np.testing.assert_allclose(dW.cpu().numpy(), ref.float().numpy(), rtol=2e-4, atol=2e-4)
    base = _rand("wbase", (N, K))
    dW2 = base.to(_dev()).clone()
    ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW2, accumulate=True)
    np.testing.assert_allclose(dW2.cpu().numpy(), (ref.float() + base).numpy(), rtol=2e-4, atol=2e-4)
    db = torch.zeros((N,), device=_dev())
    ops.colsum(dy.to(_dev()), db)
    np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().numpy(), rtol=1e-4, atol=1e-4)
    ops.colsum(dy.to(_dev()), db, accumulate=True)
    np.testing.assert_allclose(db.cpu().numpy(), 2 * dy.double().sum(0).float().numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("tokens,n_out,n_in", [(1000, 96, 192), (197 * 3, 384, 384), (130, 10, 24), (4096, 200, 72), (25, 1, 96)])
def test_linear_wgrad_fused_bias_grad(ops, tokens, n_out, n_in):
    """d2s_linear_wgrad_f32: dW and db from ONE pass over dy (split-K slabs and the single-slice path), plus accumulate."""
    g = torch.Generator().manual_seed(tokens + n_out)
    dy = torch.randn(tokens, n_out, generator=g)
    x = torch.randn(tokens, n_in, generator=g)
    dW = torch.empty(n_out, n_in, device=_dev())
    db = torch.empty(n_out, device=_dev())
    ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW, db=db)
    ref_w = (dy.double().t() @ x.double())
    ref_b = dy.double().sum(0)
    tol = 2e-6 * math.sqrt(tokens)
    assert (dW.cpu().double() - ref_w).abs().max() <= tol * ref_w.abs().max().clamp_min(1.0)
    assert (db.cpu().double() - ref_b).abs().max() <= tol * ref_b.abs().max().clamp_min(1.0)
    ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW, db=db, accumulate=True)
    assert (dW.cpu().double() - 2 * ref_w).abs().max() <= 2 * tol * ref_w.abs().max().clamp_min(1.0)
    assert (db.cpu().double() - 2 * ref_b).abs().max() <= 2 * tol * ref_b.abs().max().clamp_min(1.0)
    # without db the entry leaves nothing else behind
    dW2 = torch.empty_like(dW)
    ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW2)
    assert torch.equal(dW2.cpu() * 2, dW.cpu()) or (dW2.cpu().double() - ref_w).abs().max() <= tol * ref_w.abs().max().clamp_min(1.0)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,D,eps", [(197 * 2, 384, 1e-6), (99 * 3, 1536, 1e-5), (77, 96, 1e-5), (50, 192, 1e-6),
                                        (33, 48, 1e-5), (20, 3072, 1e-5), (10, 128, 1e-6), (9, 768, 1e-6)])
def test_layernorm_fwd_bwd(ops, rows, D, eps):
    x = _rand("lx", (rows, D), 2.0) + 0.3
    w, b = 1 + _rand("lw", (D,), 0.2), _rand("lb", (D,), 0.2)
    dy, add = _rand("ldy", (rows, D)), _rand("ladd", (rows, D))
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), wr, br, eps)
    ref.backward(dy)
    xd, wd, bd = x.to(_dev()), w.to(_dev()), b.to(_dev())
    y, mean, rstd = ops.layernorm_fwd(xd, ops.contiguous_map(rows, D), wd, bd, rows, D, eps)
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    dx = torch.empty((rows, D), device=_dev())
    dw, db = torch.zeros((D,), device=_dev()), torch.zeros((D,), device=_dev())
    ops.layernorm_bwd(xd, ops.contiguous_map(rows, D), dy.to(_dev()), wd, mean, rstd, dx, add.to(_dev()), dw, db, rows, D)
    np.testing.assert_allclose(dx.cpu().numpy(), (xr.grad + add).numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_layernorm_skip_cls_rowmap(ops):
    B, n, D = 3, 17, 128
    x = _rand("sx", (B, n, D))
    w, b = 1 + _rand("sw", (D,), 0.2), _rand("sb", (D,), 0.2)
    rows = B * (n - 1)
    ref = F.layer_norm(x[:, 1:], (D,), w, b, 1e-5).reshape(rows, D)
    xd = x.to(_dev())
    y, mean, rstd = ops.layernorm_fwd(xd, ops.skip_cls_map(n, D), w.to(_dev()), b.to(_dev()), rows, D, 1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    # backward adds into rows 1.. of a [B,n,D] gradient buffer, CLS rows untouched
    dy = _rand("sdy", (rows, D))
    gbuf = _rand("sg", (B, n, D))
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr[:, 1:], (D,), w, b, 1e-5).backward(dy.reshape(B, n - 1, D))
    gd = gbuf.to(_dev())
    ops.layernorm_bwd(xd, ops.skip_cls_map(n, D), dy.to(_dev()), w.to(_dev()), mean, rstd, gd, gd, None, None, rows, D)
    np.testing.assert_allclose(gd.cpu().numpy(), (gbuf + xr.grad).numpy(), rtol=1e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------ selection path
def test_softmax_rows(ops):
    s = _rand("sm", (37, 196), 1.5)
    got = ops.softmax_rows(s.to(_dev())).cpu()
    np.testing.assert_allclose(got.numpy(), F.softmax(s, dim=-1).numpy(), rtol=2e-6, atol=1e-9)
    s = _rand("sm2", (5, 577), 3.0)
    np.testing.assert_allclose(ops.softmax_rows(s.to(_dev())).cpu().numpy(), F.softmax(s, dim=-1).numpy(), rtol=2e-6, atol=1e-9)


def test_select_topk_bit_exact_on_reference_fixture(ops):
    g = cases.load_golden("selection")
    checked = 0
    for key in g.files:
        if not key.startswith("kept_"):
            continue
        _, N, k = key.split("_")
        N, k = int(N), int(k)
        probs = torch.from_numpy(g[f"probs_{N}"])
        kept, dropped = ops.select_topk(probs.to(_dev()), k)
        kept, dropped = kept.cpu(), dropped.cpu()
        assert kept.dtype == torch.int64
        rows = [r for r in range(probs.shape[0]) if r not in (8, 9)]
        np.testing.assert_array_equal(kept.numpy()[rows], g[key][rows])              # the reference's own output
        np.testing.assert_array_equal(dropped.numpy()[rows], g[f"dropped_{N}_{k}"][rows])
        ks, ds = O.select_topk_stable(probs, k)                                      # incl. the mass-tie rows
        np.testing.assert_array_equal(kept.numpy(), ks.numpy())
        np.testing.assert_array_equal(dropped.numpy(), ds.numpy())
        checked += 1
    assert checked >= 9


@pytest.mark.parametrize("B,n,k,D", [(4, 197, 98, 384), (3, 197, 137, 384), (2, 138, 98, 384), (5, 17, 9, 128),
                                     (2, 5, 4, 192), (2, 577, 172, 768), (3, 99, 0, 64), (2, 10, 9, 32)])
def test_gather_scatter_roundtrip(ops, B, n, k, D):
    x = _rand("gsx", (B, n, D))
    probs = torch.rand((B, n - 1), generator=torch.Generator().manual_seed(3))
    kept, _ = O.select_topk_stable(probs, k)
    ref = O.gather_pack(x, kept)
    got = ops.gather_pack(x.to(_dev()), kept.to(_dev()))
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())                    # byte moves: exact
    g = _rand("gsg", (B, k + 1, D))
    xr = x.clone().requires_grad_(True)
    O.gather_pack(xr, kept).backward(g)
    dx = ops.scatter_unpack(g.to(_dev()), kept.to(_dev()), n)
    np.testing.assert_array_equal(dx.cpu().numpy(), xr.grad.numpy())
    # encode -> decode round trip: gather(scatter(g)) == g
    np.testing.assert_array_equal(ops.gather_pack(dx, kept.to(_dev())).cpu().numpy(), g.numpy())


def test_half_mean_concat(ops):
    B, T, C = 3, 37, 512
    x = _rand("hx", (B, T, C))
    ref = torch.cat([x[:, :, :C // 2], x[:, :, C // 2:].mean(dim=1, keepdim=True).expand(B, T, C // 2)], dim=-1)
    got = ops.half_mean_concat(x.to(_dev()), B, T, C).cpu().reshape(B, T, C)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    mask = _rand("hm", (B, T, C))
    got = ops.half_mean_concat(x.to(_dev()), B, T, C, relu_mask_src=mask.to(_dev())).cpu().reshape(B, T, C)
    np.testing.assert_allclose(got.numpy(), (ref * (mask > 0)).numpy(), rtol=1e-5, atol=1e-6)
    # odd half width (small predictor of DeiT-Tiny: C = 192)
    B, T, C = 2, 9, 192
    x = _rand("hx2", (B, T, C))
    ref = torch.cat([x[:, :, :C // 2], x[:, :, C // 2:].mean(dim=1, keepdim=True).expand(B, T, C // 2)], dim=-1)
    np.testing.assert_allclose(ops.half_mean_concat(x.to(_dev()), B, T, C).cpu().reshape(B, T, C).numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_patch_embed_matches_oracle(ops):
    case = cases.MODEL_CASES["micro1"]
    cfg = case["cfg"]
    sd = {k: torch.from_numpy(v) for k, v in cases.make_weights(case)[0].items()}
    x = torch.from_numpy(cases.make_images(case))
    ref = O.embed_tokens(sd, x, cfg)
    B, T, D, P = x.shape[0], cfg["n_patches"], cfg["dim"], cfg["patch"]
    col = ops.im2col_patch(x.to(_dev()), P)
    tokens = torch.empty((B, T + 1, D), device=_dev())
    w = sd["patch_embed.proj.weight"].reshape(D, -1).contiguous().to(_dev())
    pos = sd["pos_embed"].reshape(T + 1, D).contiguous().to(_dev())
    ops.gemm(ops.NT, col, col.shape[1], w, w.shape[1], tokens, D, B * T, D, col.shape[1], ops.EPI_BIAS_ROWADD,
             sd["patch_embed.proj.bias"].to(_dev()), pos[1:], D, None, T, T, 1)
    ops.fill_cls(sd["cls_token"].reshape(D).to(_dev()), pos, tokens)
    np.testing.assert_allclose(tokens.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    g = cases.load_golden("intermediates_micro1")
    np.testing.assert_allclose(tokens.cpu().numpy(), g["tokens0"], rtol=1e-4, atol=2e-5)   # the reference's own output


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, n, H):
    q, k, v = qkv.reshape(B, n, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, n, H * 64)
    return o, a[:, :, 0, :], torch.logsumexp((q @ k.transpose(-2, -1)) * 0.125, dim=-1)


@pytest.mark.parametrize("B,n,H", [(2, 197, 6), (3, 99, 3), (2, 59, 2), (2, 17, 2), (1, 5, 3), (2, 138, 6), (1, 577, 2),
                                   (2, 32, 1), (1, 129, 1), (1, 1, 1), (1, 33, 1), (1, 40, 2), (2, 41, 1), (1, 49, 1), (1, 57, 1), (1, 72, 1)])
def test_attention_fwd_bwd(ops, B, n, H):
    qkv = _rand("aq", (B * n, 3 * H * 64), 1.0, seed=n)
    qr = qkv.clone().requires_grad_(True)
    o_ref, cls_ref, lse_ref = _attn_ref(qr, B, n, H)
    do = _rand("ado", (B, n, H * 64), 1.0, seed=n + 1)
    o_ref.backward(do)
    qd = qkv.to(_dev())
    out, lse, cls_row = ops.attn_fwd(qd, B, n, H, 0.125)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, n, -1), o_ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cls_row.cpu().numpy(), cls_ref.detach().numpy(), rtol=1e-4, atol=1e-7)
    dqkv = ops.attn_bwd(qd, out, do.reshape(B * n, -1).to(_dev()), lse, B, n, H, 0.125)
    np.testing.assert_allclose(dqkv.cpu().numpy(), qr.grad.numpy(), rtol=2e-4, atol=5e-5)


def test_softmax_with_policy_fwd_matches_reference_and_bwd_matches_autograd():
    """Row 7 of SURVEY 8a: forward against the output of the reference's own Attention.softmax_with_policy (golden), backward
    against autograd through the oracle restatement (which includes the path through the row maximum)."""
    import vit_models
    g = cases.load_golden("intermediates_micro1")
    attn, pol = torch.from_numpy(g["policy_attn_in"]), torch.from_numpy(g["policy_mask"])
    m = vit_models.Attention(128, num_heads=2, qkv_bias=True)
    ad = attn.to(_dev()).requires_grad_(True)
    out = m.softmax_with_policy(ad, pol.to(_dev()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["policy_softmax"], rtol=2e-6, atol=1e-9)
    ar = attn.clone().requires_grad_(True)
    ref = O.softmax_with_policy(ar, pol)
    go = _rand("psg", tuple(ref.shape))
    ref.backward(go)
    out.backward(go.to(_dev()))
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-4, atol=1e-6)
    # larger, with fully-kept and mostly-dropped rows
    B, H, N = 2, 3, 197
    attn = _rand("psa", (B, H, N, N), 2.0)
    pol = (torch.rand(B, N, 1, generator=torch.Generator().manual_seed(1)) > 0.6).float()
    pol[:, 0] = 1.0
    pol[1] = 1.0
    ar = attn.clone().requires_grad_(True)
    ref = O.softmax_with_policy(ar, pol)
    go = _rand("psg2", tuple(ref.shape))
    ref.backward(go)
    ad = attn.to(_dev()).requires_grad_(True)
    out = m.softmax_with_policy(ad, pol.to(_dev()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    out.backward(go.to(_dev()))
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ar.grad.numpy(), rtol=2e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ GEMM arithmetic modes
@pytest.mark.parametrize("mode,rtol,atol", [(1, 1e-4, 2e-5), (2, 3e-2, 3e-2)])
def test_gemm_split_modes(ops, mode, rtol, atol):
    """mode 1 (bf16x3 split on the bf16 matrix cores) must meet the SAME tolerance as the exact fp32 kernel and be as close to
    an fp64 product as the CPU's fp32 GEMM is; mode 2 (plain bf16 operands) only the bf16 tolerance."""
    from d2s import lib
    ops.set_gemm_mode(mode)
    try:
        for (M, N, K) in ((197 * 2, 1152, 384), (300, 96, 192), (128, 1000, 384), (391, 200, 36), (50, 1, 96), (1024, 1536, 384), (77, 192, 147)):
            x, w, b = _rand("mx", (M, K)), _rand("mw", (N, K), 0.05), _rand("mb", (N,), 0.1)
            ref64 = F.linear(x.double(), w.double(), b.double())
            ref32 = F.linear(x, w, b)
            got = ops.linear_fwd(x.to(_dev()), w.to(_dev()), b.to(_dev())).cpu()
            np.testing.assert_allclose(got.numpy(), ref32.numpy(), rtol=rtol, atol=atol)
            if mode == 1:
                e_hip = float((got.double() - ref64).norm() / ref64.norm())
                e_cpu = float((ref32.double() - ref64).norm() / ref64.norm())
                assert e_hip <= max(4 * e_cpu, 3e-7), (M, N, K, e_hip, e_cpu)
            # dgrad goes through the transposed-weight NT path
            dy = _rand("mdy", (M, N))
            dx = ops.linear_dgrad(dy.to(_dev()), w.to(_dev())).cpu()
            np.testing.assert_allclose(dx.numpy(), (dy @ w).numpy(), rtol=rtol, atol=atol * 10)
            r = _rand("mr", (M, N))
            got = ops.linear_fwd(x.to(_dev()), w.to(_dev()), b.to(_dev()), epi=ops.EPI_BIAS_RESID, aux=r.to(_dev())).cpu()
            np.testing.assert_allclose(got.numpy(), (ref32 + r).numpy(), rtol=rtol, atol=atol)
    finally:
        ops.set_gemm_mode(0)


@pytest.mark.parametrize("mode,rtol,atol", [(1, 1e-4, 2e-5), (2, 3e-2, 3e-2)])
@pytest.mark.parametrize("M,N,K", [(4096, 1536, 384), (2500, 384, 1536), (2048, 768, 96), (3000, 1152, 384), (2304, 300, 64)])
def test_gemm_split_modes_big_tiles(ops, mode, rtol, atol, M, N, K):
    """Shapes that take the large-tile LDS-DMA kernel in bf16 mode (M >= 2048; 256x256 or 256x128 tiles by shape, the GPU round also
    runs this file with D2S_SPLIT_DMA=2 / 3 to force either): forward with every epilogue the model uses on that path (bias, GELU +
    pre-activation copy, residual, GELU-gradient mask) and the input-gradient layout, ragged edges in M and N included, in both
    arithmetic modes at the tolerances of the 128x128 kernel."""
    ops.set_gemm_mode(mode)
    try:
        x, w, b = _rand("bx", (M, K), seed=M), _rand("bw", (N, K), 0.05, seed=N), _rand("bb", (N,), 0.1)
        r = _rand("br", (M, N), seed=K)
        ref = F.linear(x, w, b)
        xd, wd, bd, rd = x.to(_dev()), w.to(_dev()), b.to(_dev()), r.to(_dev())
        np.testing.assert_allclose(ops.linear_fwd(xd, wd, bd).cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol)
        z = torch.empty((M, N), device=_dev())
        h = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_GELU, aux_out=z)
        np.testing.assert_allclose(z.cpu().numpy(), ref.numpy(), rtol=rtol, atol=atol)
        np.testing.assert_allclose(h.cpu().numpy(), F.gelu(ref).numpy(), rtol=rtol, atol=atol)
        np.testing.assert_allclose(ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_RESID, aux=rd).cpu().numpy(), (ref + r).numpy(), rtol=rtol, atol=atol)
        dy = _rand("bdy", (M, N), seed=M + N)
        dref = dy @ w
        np.testing.assert_allclose(ops.linear_dgrad(dy.to(_dev()), wd).cpu().numpy(), dref.numpy(), rtol=rtol, atol=atol * 10)
        if K % 4 == 0:
            zz = _rand("bz", (M, K), seed=7)
            zt = zz.clone().requires_grad_(True)
            F.gelu(zt).backward(dref)
            got = ops.linear_dgrad(dy.to(_dev()), wd, epi=ops.EPI_MUL_GELU_GRAD, aux=zz.to(_dev())).cpu()
            np.testing.assert_allclose(got.numpy(), zt.grad.numpy(), rtol=rtol, atol=atol * 10)
    finally:
        ops.set_gemm_mode(0)


@pytest.mark.parametrize("M,N,K", [(4096, 1536, 384), (2500, 384, 1536), (394, 1152, 384), (3000, 768, 768), (2304, 3072, 768)])
def test_gemm_bf16_io(ops, M, N, K):
    """d2s_gemm_f32_bf16io: with a16 = bf16(x) the result is the product of the bf16-rounded operands accumulated in fp32 (products of
    bf16 numbers are exact in fp32, so only the summation order separates it from a float64 product of the same rounded operands:
    rtol 2e-5); c16 is bit-for-bit bf16(C); A / C may be omitted when their bf16 forms are given; NT and NN layouts, GELU epilogue with
    its pre-activation copy."""
    ops.set_gemm_mode(2)
    try:
        x, w, b = _rand("ix", (M, K), seed=M), _rand("iw", (N, K), 0.05, seed=N), _rand("ib", (N,), 0.1)
        xr, wr = x.bfloat16().float(), w.bfloat16().float()
        ref = (xr.double() @ wr.double().t() + b.double()).float()
        xd, wd, bd = x.to(_dev()), w.to(_dev()), b.to(_dev())
        x16 = xd.bfloat16()
        c16 = torch.empty((M, N), dtype=torch.bfloat16, device=_dev())
        got = ops.linear_fwd(xd, wd, bd, a16=x16, c16=c16)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
        assert torch.equal(c16, got.bfloat16())
        # no fp32 A, no fp32 C
        c16b = torch.empty_like(c16)
        assert ops.linear_fwd(None, wd, bd, a16=x16, c16=c16b, want_f32=False) is None
        assert torch.equal(c16b, c16)
        # GELU epilogue: pre-activation copy fp32, bf16 copy of the activation
        z = torch.empty((M, N), device=_dev())
        h = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_GELU, aux_out=z, a16=x16, c16=c16)
        np.testing.assert_allclose(z.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(h.cpu().numpy(), F.gelu(ref).numpy(), rtol=1e-4, atol=2e-5)
        assert torch.equal(c16, h.bfloat16())
        # the pre-activation kept in bf16 (epilogue codes 9 / 10, picked from the tensor's dtype): the same values rounded once, the same
        # activation; fc2's input gradient reading it back equals the fp32-aux call on the rounded values
        z16 = torch.empty((M, N), dtype=torch.bfloat16, device=_dev())
        h2 = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_BIAS_GELU, aux_out=z16, a16=x16, c16=c16b)
        assert torch.equal(z16, z.bfloat16()) and torch.equal(h2, h) and torch.equal(c16b, c16)
        if N % 32 == 0:
            gyd = _rand("igy", (M, K), seed=M + 7).to(_dev())
            w2 = _rand("iw2", (K, N), 0.05, seed=K).to(_dev())                    # fc2.weight [D, hidden]: dz = (gy @ w2) * gelu'(z)
            dz_a = ops.linear_dgrad(gyd, w2, epi=ops.EPI_MUL_GELU_GRAD, aux=z16.float(), a16=gyd.bfloat16())
            dz_b = ops.linear_dgrad(gyd, w2, epi=ops.EPI_MUL_GELU_GRAD, aux=z16, a16=gyd.bfloat16())
            assert torch.equal(dz_a, dz_b)
        # the weight handed over in bf16 (d2s_convert_bf16 once, instead of a conversion inside every call): identical result
        from d2s import lib
        w16 = torch.empty((N, K), dtype=torch.bfloat16, device=_dev())
        lib.call("d2s_convert_bf16", lib.ptr(wd), lib.ptr(w16), wd.numel())
        assert torch.equal(w16, wd.bfloat16())
        out_b = torch.empty((M, N), device=_dev())
        ops.gemm(ops.NT, xd, K, None, K, out_b, N, M, N, K, ops.EPI_BIAS, bd, a16=x16, b16=w16)
        assert torch.equal(out_b, got)
        wt16 = wd.t().contiguous().bfloat16()                     # [K][N]: the k-contiguous B operand of dx = dy W
        dxb = torch.empty((M, K), device=_dev())
        dyd0 = _rand("idy", (M, N), seed=M + N).to(_dev())
        ops.gemm(ops.NN, dyd0, N, None, K, dxb, K, M, K, N, ops.EPI_NONE, None, b16=wt16)
        np.testing.assert_allclose(dxb.cpu().numpy(), (dyd0.cpu().bfloat16().double() @ wr.double()).float().numpy(), rtol=2e-5, atol=2e-4)
        # input-gradient layout: dx = dy @ W with a16 = bf16(dy) and a bf16 copy of dx
        dy = _rand("idy", (M, N), seed=M + N)
        dref = (dy.bfloat16().double() @ wr.double()).float()
        dyd = dy.to(_dev())
        d16 = torch.empty((M, K), dtype=torch.bfloat16, device=_dev())
        dx = ops.linear_dgrad(dyd, wd, a16=dyd.bfloat16(), c16=d16)
        np.testing.assert_allclose(dx.cpu().numpy(), dref.numpy(), rtol=2e-5, atol=2e-4)
        assert torch.equal(d16, dx.bfloat16())
    finally:
        ops.set_gemm_mode(0)


@pytest.mark.parametrize("M,N,K", [(8192, 1536, 768), (36928, 768, 768), (8192, 384, 768), (5000, 768, 3072)])      # 256x256, 256x256, 256x128, 256x128 tiles
def test_gemm_bf16_dma_race_screen(ops, M, N, K):
    """The LDS-DMA kernel orders its LDS traffic by hand (counted vmcnt waits + raw barriers, persistent workgroups).  The summation
    order is fixed, so repeated launches must be bit-identical; a read that runs ahead of its DMA, or a DMA that overwrites a buffer still
    being read, shows up as a launch that differs.  40 launches per shape (the first checked against a float64 product of the rounded
    operands), on both tile shapes and with other work keeping the memory system busy in between."""
    ops.set_gemm_mode(2)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.05
        b = torch.randn(N, generator=g)
        xd, wd, bd = x.to(_dev()), w.to(_dev()), b.to(_dev())
        x16 = xd.bfloat16()
        noise = torch.empty(64 << 20, device=_dev())
        first = ops.linear_fwd(xd, wd, bd, a16=x16).clone()
        rows = torch.randint(0, M, (256,), generator=g)
        ref = (x[rows].bfloat16().double() @ w.bfloat16().double().t() + b.double()).float()
        np.testing.assert_allclose(first[rows.to(_dev())].cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
        for i in range(40):
            if i % 3 == 0:
                noise.normal_()                 # evict / load the memory system between launches
            out = ops.linear_fwd(xd, wd, bd, a16=x16)
            assert torch.equal(out, first), f"launch {i} differs from the first: max |diff| {float((out - first).abs().max()):.3e}"
    finally:
        ops.set_gemm_mode(0)


def test_bf16_io_rejected_outside_bf16_mode(ops):
    """The bf16 side channels exist in mode 2 only: every other use is an argument error, not a silent fp32 path."""
    from d2s import lib
    x, w = _rand("rx", (256, 64)).to(_dev()), _rand("rw", (128, 64)).to(_dev())
    out = torch.empty((256, 128), device=_dev())
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=_dev())
    # K % 32 != 0 for a16
    x2, w2 = _rand("rx2", (256, 40)).to(_dev()), _rand("rw2", (128, 40)).to(_dev())
    with pytest.raises(RuntimeError):
        lib.call("d2s_gemm_f32_bf16io", 0, lib.ptr(x2), 40, lib.ptr(w2), 40, lib.ptr(out), 128, 256, 128, 40, 0, None, None, 0, None,
                 lib.ptr(x2.bfloat16()), None, None, lib.ptr(ws), ws.numel())
    # the weight-gradient layout has no bf16 side channel
    with pytest.raises(RuntimeError):
        lib.call("d2s_gemm_f32_bf16io", 2, lib.ptr(x), 64, lib.ptr(w), 64, lib.ptr(out), 128, 256, 128, 64, 0, None, None, 0, None,
                 lib.ptr(x.bfloat16()), None, None, lib.ptr(ws), ws.numel())


@pytest.mark.parametrize("rows,D", [(197 * 3, 384), (1000, 768), (77, 192), (130, 1536)])
def test_layernorm_fwd_bf16out(ops, rows, D):
    """d2s_layernorm_fwd_bf16out: fp32 output identical to d2s_layernorm_fwd, the bf16 copy is its rounding, and the bf16-only form
    (no fp32 output, no statistics) writes the same bf16 values."""
    x, w, b = _rand("lx", (rows, D)).to(_dev()), _rand("lw", (D,), 0.5).to(_dev()) + 1, _rand("lb", (D,), 0.1).to(_dev())
    cmap = ops.contiguous_map(rows, D)
    y, mean, rstd = ops.layernorm_fwd(x, cmap, w, b, rows, D, 1e-6)
    y2, mean2, rstd2, y16 = ops.layernorm_fwd_bf16(x, cmap, w, b, rows, D, 1e-6)
    assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert torch.equal(y16, y.bfloat16())
    y3, m3, r3, y16b = ops.layernorm_fwd_bf16(x, cmap, w, b, rows, D, 1e-6, stats=False, want_f32=False)
    assert y3 is None and m3 is None and r3 is None and torch.equal(y16b, y16)


def test_layernorm_bwd_bf16out(ops):
    """d2s_layernorm_bwd_bf16out: identical fp32 outputs, bf16 copy = rounding of dx (after the residual add)."""
    rows, D = 1000, 768
    x, dy, w = _rand("bx", (rows, D)).to(_dev()), _rand("bdy", (rows, D)).to(_dev()), (_rand("bw", (D,), 0.5) + 1).to(_dev())
    add = _rand("badd", (rows, D)).to(_dev())
    cmap = ops.contiguous_map(rows, D)
    _, mean, rstd = ops.layernorm_fwd(x, cmap, w, torch.zeros(D, device=_dev()), rows, D, 1e-6)
    dx1, dw1, db1 = torch.empty_like(x), torch.empty(D, device=_dev()), torch.empty(D, device=_dev())
    ops.layernorm_bwd(x, cmap, dy, w, mean, rstd, dx1, add, dw1, db1, rows, D)
    dx2, dw2, db2 = torch.empty_like(x), torch.empty(D, device=_dev()), torch.empty(D, device=_dev())
    dx16 = torch.empty((rows, D), dtype=torch.bfloat16, device=_dev())
    ops.layernorm_bwd(x, cmap, dy, w, mean, rstd, dx2, add, dw2, db2, rows, D, dx16=dx16)
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2) and torch.equal(db1, db2)
    assert torch.equal(dx16, dx1.bfloat16())


def test_attn_bwd_bf16out(ops):
    """d2s_attn_bwd_bf16_bf16out: same dqkv as d2s_attn_bwd_bf16 and its bf16 rounding."""
    B, n, H = 2, 197, 6
    qkv = _rand("bq", (B * n, 3 * H * 64), 0.5).to(_dev())
    dout = _rand("bdo", (B * n, H * 64), 0.5).to(_dev())
    ops.set_gemm_mode(2)
    try:
        out, lse, _ = ops.attn_fwd(qkv, B, n, H, 0.125, False)
        d1 = ops.attn_bwd(qkv, out, dout, lse, B, n, H, 0.125)
        d16 = torch.empty(qkv.shape, dtype=torch.bfloat16, device=_dev())
        d2 = ops.attn_bwd(qkv, out, dout, lse, B, n, H, 0.125, dqkv16=d16)
        assert torch.equal(d1, d2) and torch.equal(d16, d1.bfloat16())
        d16b = torch.empty_like(d16)
        d3 = ops.attn_bwd(qkv.bfloat16(), out, dout, lse, B, n, H, 0.125, dqkv16=d16b)      # bf16 qkv in: identical gradients
        assert torch.equal(d3, d1) and torch.equal(d16b, d16)
        d16c = torch.empty_like(d16)
        assert ops.attn_bwd(qkv, out, dout, lse, B, n, H, 0.125, dqkv16=d16c, want_f32=False) is None and torch.equal(d16c, d16)
    finally:
        ops.set_gemm_mode(0)


def test_attn_fwd_bf16out(ops):
    """d2s_attn_fwd_bf16_bf16out: same fp32 outputs as d2s_attn_fwd_bf16, bf16 copy = rounding of the output, and the bf16-only form."""
    B, n, H = 3, 197, 6
    qkv = _rand("aq", (B * n, 3 * H * 64), 0.5).to(_dev())
    ops.set_gemm_mode(2)
    try:
        out, lse, cls_row = ops.attn_fwd(qkv, B, n, H, 0.125, True)
        out2, lse2, cls2, o16 = ops.attn_fwd_bf16io(qkv, B, n, H, 0.125, True)
        assert torch.equal(out, out2) and torch.equal(lse, lse2) and torch.equal(cls_row, cls2)
        assert torch.equal(o16, out.bfloat16())
        out3, _, _, o16b = ops.attn_fwd_bf16io(qkv, B, n, H, 0.125, False, want_f32=False)
        assert out3 is None and torch.equal(o16b, o16)
        # qkv handed over in bf16 (the qkv GEMM's bf16 copy): the kernel rounds to the same numbers itself -> identical results
        q16 = qkv.bfloat16()
        out4, lse4, cls4, o16c = ops.attn_fwd_bf16io(q16, B, n, H, 0.125, True)
        assert torch.equal(out4, out) and torch.equal(lse4, lse) and torch.equal(cls4, cls_row) and torch.equal(o16c, o16)
    finally:
        ops.set_gemm_mode(0)


@pytest.mark.parametrize("tokens,n_out,n_in", [(197 * 8, 384, 1536), (1000, 96, 192), (130, 10, 24), (4100, 200, 72),
                                               (5000, 768, 768), (9991, 512, 1024), (2100, 3072, 768)])
def test_linear_wgrad_bf16_mode(ops, tokens, n_out, n_in):
    """Mode 2 runs the weight gradient on the bf16 matrix cores (transposing split + K-sliced matrix kernel - the 256x256 LDS-DMA kernel
    where the weight fills its tiles, last three shapes - + ordered combine); the bias gradient stays an exact fp32 column sum.
    Tolerance: bf16 operand rounding (2^-9 relative per product)."""
    g = torch.Generator().manual_seed(tokens + n_in)
    dy = torch.randn(tokens, n_out, generator=g)
    x = torch.randn(tokens, n_in, generator=g)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    ops.set_gemm_mode(ops.GEMM_BF16)
    try:
        dW = torch.empty(n_out, n_in, device=_dev())
        db = torch.empty(n_out, device=_dev())
        ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW, db=db)
        scale = math.sqrt(tokens)          # rms magnitude of an entry of dy^T x
        assert (dW.cpu().double() - ref_w).abs().max() <= 0.02 * scale
        assert (dW.cpu().double() - ref_w).pow(2).mean().sqrt() <= 0.004 * scale
        assert (db.cpu().double() - ref_b).abs().max() <= 2e-6 * scale * 10
        ops.linear_wgrad(dy.to(_dev()), x.to(_dev()), dW, db=db, accumulate=True)
        assert (dW.cpu().double() - 2 * ref_w).abs().max() <= 0.04 * scale
        assert (db.cpu().double() - 2 * ref_b).abs().max() <= 4e-6 * scale * 10
    finally:
        ops.set_gemm_mode(ops.GEMM_EXACT)


@pytest.mark.parametrize("tokens,n_out,n_in", [(197 * 8, 384, 1536), (5000, 768, 768), (1001, 96, 200),
                                                (4096, 768, 1024), (2304, 1160, 768), (173 * 64, 2304, 768), (8192, 512, 264)])
def test_linear_wgrad_bf16_input(ops, tokens, n_out, n_in):
    """d2s_linear_wgrad_f32_bf16x: with the layer input handed over in bf16 (what the bf16 data path saves) the weight and bias gradients
    are bit-identical to the fp32-input call, which rounds the same values itself.  The last four shapes (tokens a multiple of 64, outputs
    that fill 256x256 tiles) take the token-major matrix kernel when both operands come in bf16 (no transposing pass; csrc/gemm_split.hip
    gemm_bf16_dma_kernel<.., TOK>): same products in the same order, so still bit-identical - ragged tile edges included."""
    g = torch.Generator().manual_seed(tokens)
    dy = torch.randn(tokens, n_out, generator=g).to(_dev())
    x = torch.randn(tokens, n_in, generator=g).to(_dev())
    ops.set_gemm_mode(ops.GEMM_BF16)
    try:
        dW1, db1 = torch.empty(n_out, n_in, device=_dev()), torch.empty(n_out, device=_dev())
        ops.linear_wgrad(dy, x, dW1, db=db1)
        dW2, db2 = torch.empty(n_out, n_in, device=_dev()), torch.empty(n_out, device=_dev())
        ops.linear_wgrad(dy, None, dW2, db=db2, x16=x.bfloat16())
        assert torch.equal(dW1, dW2) and torch.equal(db1, db2)
        # the gradient in bf16 as well: same weight gradient (the same rounded values are multiplied); the bias gradient is the sum of the
        # bf16 values, 2^-9 relative per element away from the exact column sum
        dW3, db3 = torch.empty(n_out, n_in, device=_dev()), torch.empty(n_out, device=_dev())
        ops.linear_wgrad(None, None, dW3, db=db3, x16=x.bfloat16(), dy16=dy.bfloat16())
        assert torch.equal(dW3, dW1)
        ref_b = dy.bfloat16().double().sum(0)
        assert (db3.double() - ref_b).abs().max() <= 1e-5 * math.sqrt(tokens) * 10
        # both forms of the gradient (fp32 + its bf16 rounding, as the LayerNorm backward hands them over): weights from the bf16 operands,
        # bias from the fp32 one - bit-identical to the fp32-only call
        dW4, db4 = torch.empty(n_out, n_in, device=_dev()), torch.empty(n_out, device=_dev())
        ops.linear_wgrad(dy, None, dW4, db=db4, x16=x.bfloat16(), dy16=dy.bfloat16())
        assert torch.equal(dW4, dW1) and torch.equal(db4, db1)
    finally:
        ops.set_gemm_mode(ops.GEMM_EXACT)


@pytest.mark.parametrize("M,N,K", [(600, 384, 1536), (3168, 384, 1536), (257, 96, 2048), (1000, 200, 1024)])
def test_gemm_small_grid_split_k_epilogues(ops, M, N, K):
    """Forward / dgrad GEMMs with a small tile grid and a long K split K and apply their epilogue in the ordered slab combine
    (splitk_reduce_epi_kernel): every epilogue kind must match the single-pass formula."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 0.5
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    need = ops.lib.query("d2s_gemm_f32_workspace_bytes", 0, M, N, K, ops.GEMM_EXACT)
    assert need > 0, "this shape is expected to take the split-K path"
    ref = (x.double() @ w.double().t())
    d = _dev()
    tol = dict(rtol=2e-4, atol=2e-4)
    y = ops.linear_fwd(x.to(d), w.to(d), b.to(d)).cpu()
    np.testing.assert_allclose(y.numpy(), (ref + b.double()).float().numpy(), **tol)
    y = ops.linear_fwd(x.to(d), w.to(d), b.to(d), epi=ops.EPI_BIAS_RELU).cpu()
    np.testing.assert_allclose(y.numpy(), torch.relu(ref + b.double()).float().numpy(), **tol)
    pre = torch.empty(M, N, device=d)
    y = ops.linear_fwd(x.to(d), w.to(d), b.to(d), epi=ops.EPI_BIAS_GELU, aux_out=pre).cpu()
    np.testing.assert_allclose(pre.cpu().numpy(), (ref + b.double()).float().numpy(), **tol)
    np.testing.assert_allclose(y.numpy(), torch.nn.functional.gelu(ref + b.double()).float().numpy(), **tol)
    y = ops.linear_fwd(x.to(d), w.to(d), b.to(d), epi=ops.EPI_BIAS_RESID, aux=r.to(d)).cpu()
    np.testing.assert_allclose(y.numpy(), (ref + b.double() + r.double()).float().numpy(), **tol)
    # dgrad layout with the activation-gradient masks
    dy = torch.randn(M, N, generator=g) * 0.5
    w2 = torch.randn(N, K, generator=g) * 0.05          # dx[M,K] = dy[M,N] @ w2[N,K]: reduction length N - use the transposed problem
    z = torch.randn(M, N, generator=g)
    dyk = torch.randn(M, K, generator=g) * 0.5
    wk = torch.randn(K, N, generator=g) * 0.05           # dx[M,N] = dyk[M,K] @ wk[K,N]: reduction length K (long)
    refd = dyk.double() @ wk.double()
    got = ops.linear_dgrad(dyk.to(d), wk.to(d), epi=ops.EPI_MUL_RELU_MASK, aux=z.to(d)).cpu()
    np.testing.assert_allclose(got.numpy(), (refd * (z > 0)).float().numpy(), **tol)
    zz = z.clone().double().requires_grad_(True)
    torch.nn.functional.gelu(zz).backward(torch.ones_like(zz))
    got = ops.linear_dgrad(dyk.to(d), wk.to(d), epi=ops.EPI_MUL_GELU_GRAD, aux=z.to(d)).cpu()
    np.testing.assert_allclose(got.numpy(), (refd * zz.grad).float().numpy(), **tol)


@pytest.mark.parametrize("mode", ["exact", "split", "bf16"])
def test_gemm_fuzz_shapes_all_layouts(ops, mode):
    """Seeded sweep over awkward shapes (edges that are not tile multiples, tiny N / M, K a multiple of 16 -> guard-free FAST loop with
    clamped rows; K not a multiple of 16 or unaligned leading dimensions -> guarded loop) for the three layouts, against fp64."""
    rng = np.random.default_rng(1234)
    d = _dev()
    ops.set_gemm_mode({"exact": ops.GEMM_EXACT, "split": ops.GEMM_SPLIT, "bf16": ops.GEMM_BF16}[mode])
    try:
        _gemm_fuzz_body(ops, rng, d, mode)
    finally:
        ops.set_gemm_mode(ops.GEMM_EXACT)


def _gemm_fuzz_body(ops, rng, d, mode):
    for case in range(36):
        M = int(rng.choice([1, 3, 4, 5, 31, 33, 64, 65, 127, 129, 200, 257, 300]))
        N = int(rng.choice([1, 2, 4, 7, 8, 36, 63, 64, 65, 100, 129, 192]))
        K = int(rng.choice([16, 32, 48, 64, 96, 112, 160, 20, 50, 147]))
        layout = case % 3
        g = torch.Generator().manual_seed(case)
        a = torch.randn(M, K, generator=g)
        b = torch.randn(N, K, generator=g)
        ref = (a.double() @ b.double().t()).float().numpy()
        tol = dict(rtol=2e-4, atol=2e-4) if mode != "bf16" else dict(rtol=3e-2, atol=0.05 * math.sqrt(K))
        if layout == 0:       # NT: y = a @ b^T
            bias = torch.randn(N, generator=g)
            got = ops.linear_fwd(a.to(d), b.to(d), bias.to(d)).cpu().numpy()
            np.testing.assert_allclose(got, ref + bias.numpy(), err_msg=f"NT {M}x{N}x{K}", **tol)
        elif layout == 1:     # NN: dx[M,N] = a[M,K] @ w[K,N]
            w = b.t().contiguous()
            got = ops.linear_dgrad(a.to(d), w.to(d)).cpu().numpy()
            np.testing.assert_allclose(got, ref, err_msg=f"NN {M}x{N}x{K}", **tol)
        else:                 # TN: dW[M,N] = dy[K,M]^T @ x[K,N]  (+ fused bias gradient)
            dy, x = a.t().contiguous(), b.t().contiguous()
            dW = torch.empty(M, N, device=d)
            db = torch.empty(M, device=d)
            ops.linear_wgrad(dy.to(d), x.to(d), dW, db=db)
            np.testing.assert_allclose(dW.cpu().numpy(), ref, err_msg=f"TN {M}x{N}x{K}", **tol)
            np.testing.assert_allclose(db.cpu().numpy(), dy.double().sum(0).float().numpy(), rtol=1e-4, atol=1e-4, err_msg=f"TN bias {M}x{K}")


@pytest.mark.parametrize("R,C,training", [(4 * 16, 128, True), (128 * 5, 96, True), (777, 384, True), (300, 64, False), (2 * 196, 1536, True)])
def test_batchnorm_fwd_bwd(ops, R, C, training):
    """d2s_batchnorm_fwd / bwd (the --predictor-bn BatchNormLayer, dynamic_vit.py:350-367) against torch's batch_norm on the same rows:
    output, saved statistics, running estimates after the update, dx (with and without the ReLU-output mask), dw, db."""
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g) * 1.7 + 0.4
    w = 1 + 0.3 * torch.randn(C, generator=g)
    b = 0.2 * torch.randn(C, generator=g)
    dy = torch.randn(R, C, generator=g)
    rm0, rv0 = 0.1 * torch.randn(C, generator=g), 1 + 0.2 * torch.rand(C, generator=g)
    xr = x.clone().double().requires_grad_(True)
    wr, br = w.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
    rm, rv = rm0.clone().double(), rv0.clone().double()
    yr = torch.nn.functional.batch_norm(xr, rm, rv, wr, br, training=training, momentum=0.1, eps=1e-5)
    yr.backward(dy.double())
    d = _dev()
    rmd, rvd = rm0.to(d), rv0.to(d)
    y, mean, rstd = ops.batchnorm_fwd(x.to(d), w.to(d), b.to(d), rmd, rvd, training)
    np.testing.assert_allclose(y.cpu().numpy(), yr.detach().float().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.float().numpy(), rtol=1e-5, atol=1e-6)
    dw, db = torch.empty(C, device=d), torch.empty(C, device=d)
    dx = ops.batchnorm_bwd(x.to(d), dy.to(d), w.to(d), mean, rstd, dw, db, training)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.float().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.float().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.float().numpy(), rtol=2e-4, atol=2e-4)
    dxm = ops.batchnorm_bwd(x.to(d), dy.to(d), w.to(d), mean, rstd, None, None, training, relu_mask=True)
    np.testing.assert_allclose(dxm.cpu().numpy(), (xr.grad.float() * (x > 0)).numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("B,n,H", [(2, 197, 6), (3, 99, 3), (1, 577, 2), (2, 32, 1), (1, 33, 1), (1, 1, 1), (2, 41, 2), (1, 130, 1)])
def test_attention_fwd_bf16_mode(ops, B, n, H):
    """d2s_attn_fwd_bf16 (bf16 arithmetic mode): output, log-sum-exp and CLS row against the fp32 reference within bf16 operand rounding
    (2^-9 relative on q, k, v), and the fp32 backward run from ITS outputs stays close to autograd's gradients."""
    qkv = _rand("aqb", (B * n, 3 * H * 64), 1.0, seed=n)
    qr = qkv.clone().requires_grad_(True)
    o_ref, cls_ref, lse_ref = _attn_ref(qr, B, n, H)
    do = _rand("adob", (B, n, H * 64), 1.0, seed=n + 1)
    o_ref.backward(do)
    qd = qkv.to(_dev())
    ops.set_gemm_mode(ops.GEMM_BF16)
    try:
        out, lse, cls_row = ops.attn_fwd(qd, B, n, H, 0.125)
        dqkv = ops.attn_bwd(qd, out, do.reshape(B * n, -1).to(_dev()), lse, B, n, H, 0.125)
    finally:
        ops.set_gemm_mode(ops.GEMM_EXACT)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, n, -1), o_ref.detach().numpy(), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.detach().numpy(), rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(cls_row.cpu().numpy(), cls_ref.detach().numpy(), rtol=0.1, atol=2e-3)
    np.testing.assert_allclose(cls_row.cpu().numpy().sum(-1), np.ones((B, H), np.float32), rtol=1e-4)
    err = (dqkv.cpu() - qr.grad).norm() / qr.grad.norm()
    assert float(err) < 3e-2, float(err)


@pytest.mark.parametrize("mode", ["2", "0"])
def test_bf16_attention_forward_both_tile_forms(mode):
    """The bf16 attention forward exists in two tile forms - 32-key tiles with a transposed V image, and 64-key tiles with a row-major V image
    read through ds_read_b64_tr_b16 - chosen by sequence length (D2S_ATTN_BF16_T64, read once per process).  Every bf16 attention test of
    this file must pass with either form forced for every length."""
    import subprocess
    import sys
    env = dict(os.environ, D2S_ATTN_BF16_T64=mode)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(cases.REPO, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-q", "-x",
                          "-k", "(attn or attention) and bf16 and not both_tile_forms"], env=env, capture_output=True, text=True, timeout=900, cwd=cases.REPO)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout
