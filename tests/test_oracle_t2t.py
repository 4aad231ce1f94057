"""CPU tier: the oracle's T2T restatement (t2t_vit.py / token_performer.py / token_transformer.py / transformer_block.py)
against fixtures produced by the reference's own classes (tools/gen_golden.py::gen_t2t)."""
import numpy as np
import pytest
import torch

from tests import cases
from oracle import d2s_oracle as O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("tt", ["performer", "transformer"])
def test_t2t_matches_reference(tt):
    g = cases.load_golden("t2t")
    c = cases.T2T_CASE
    sd = {k: _t(v).requires_grad_(v.dtype == np.float32) for k, v in cases.make_t2t_weights(tt).items()}
    x = _t(cases.make_t2t_images())
    tok0 = O.unfold_tokens(x, 7, 4, 2)
    np.testing.assert_allclose(tok0.numpy(), g[f"{tt}_unfold0"], rtol=0, atol=0)
    stage = O.token_performer if tt == "performer" else O.token_transformer
    a1 = stage(sd, "tokens_to_token.attention1.", tok0)
    np.testing.assert_allclose(a1.detach().numpy(), g[f"{tt}_attention1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.t2t_module(sd, x, tt).detach().numpy(), g[f"{tt}_t2t_module"], rtol=1e-5, atol=1e-6)
    logits, heads = O.t2t_forward(sd, x, c["depth"], c["heads"], tt)
    np.testing.assert_allclose(logits.detach().numpy(), g[f"{tt}_logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(heads[-1].detach().numpy(), g[f"{tt}_block_head_last"], rtol=1e-5, atol=1e-5)
    from d2s import synth
    gl = _t(synth.normal("t2t/g", tuple(logits.shape), seed=9))
    (logits * gl).sum().backward()
    for n, ref in zip([str(s) for s in g[f"{tt}_grad_names"]], g[f"{tt}_grad_norms"]):
        if ref < 0:        # frozen: pos_embed and the performer's random features w
            assert sd[n].grad is None or n.endswith(".w") or n == "pos_embed", n
            continue
        np.testing.assert_allclose(float(sd[n].grad.double().norm()), ref, rtol=2e-4, atol=1e-9, err_msg=n)


def test_sinusoid_matches_reference_table():
    # pos_embed of the reference model is the sinusoid table itself (frozen), so the state dict built by cases must equal it
    sd = cases.make_t2t_weights("performer")
    t = O.sinusoid_encoding(17, 128).numpy()
    np.testing.assert_array_equal(sd["pos_embed"], t)
    assert abs(float(t[0, 1, 0]) - np.sin(1.0)) < 1e-6 and abs(float(t[0, 0, 1]) - 1.0) < 1e-7


@pytest.mark.parametrize("tt", ["performer", "transformer"])
def test_t2t_vit_14_at_224_matches_reference(tt):
    """BASELINE config 4 at its own geometry (T2T-ViT-14, one 224x224 image): the oracle against the fixture the reference's T2T_ViT
    produced - 3136-token soft split, first token encoder, T2T module, logits, last block head, every gradient norm."""
    from d2s import synth
    g = cases.load_golden("t2t_224")
    c = cases.T2T_224_CASE
    sd = {k: _t(v).requires_grad_(v.dtype == np.float32) for k, v in cases.make_t2t_weights(tt, case=c).items()}
    x = _t(cases.make_t2t_images(c))
    tok0 = O.unfold_tokens(x, 7, 4, 2)
    assert list(tok0.shape) == g[f"{tt}_unfold0_shape"].tolist() == [1, 3136, 147]
    np.testing.assert_array_equal(tok0[:, 1000:1004].numpy(), g[f"{tt}_unfold0_slice"])
    stage = O.token_performer if tt == "performer" else O.token_transformer
    a1 = stage(sd, "tokens_to_token.attention1.", tok0)
    np.testing.assert_allclose(a1[:, ::392].detach().numpy(), g[f"{tt}_attention1_slice"], rtol=1e-4, atol=2e-6)
    tm = O.t2t_module(sd, x, tt)
    assert list(tm.shape) == g[f"{tt}_t2t_module_shape"].tolist() == [1, 196, 384]
    np.testing.assert_allclose(tm[:, ::28, ::8].detach().numpy(), g[f"{tt}_t2t_module_slice"], rtol=1e-4, atol=2e-6)
    logits, heads = O.t2t_forward(sd, x, c["depth"], c["heads"], tt)
    assert len(heads) == int(g[f"{tt}_n_block_heads"])
    np.testing.assert_allclose(logits.detach().numpy(), g[f"{tt}_logits"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(heads[-1][:, ::16].detach().numpy(), g[f"{tt}_block_head_last_slice"], rtol=1e-4, atol=1e-5)
    gl = _t(synth.normal("t2t224/g", tuple(logits.shape), seed=9))
    (logits * gl).sum().backward()
    for n, ref in zip([str(s) for s in g[f"{tt}_grad_names"]], g[f"{tt}_grad_norms"]):
        if ref < 0:
            continue
        np.testing.assert_allclose(float(sd[n].grad.double().norm()), ref, rtol=1e-3, atol=1e-9, err_msg=n)
