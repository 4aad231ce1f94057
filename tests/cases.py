"""Shared definitions of the parity cases: geometry, batch, seeds.  Used by tools/gen_golden.py (which
runs the reference on them) and by the tests (which run the oracle and the HIP path on them)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "dense2sparse-vit_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

from d2s import synth  # noqa: E402
from oracle import d2s_oracle as O  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")

# name -> dict(cfg, batch, seed).  Ratios for the micro geometry are small because the reference computes
# the keep count from the hard-coded init_n = 196 (dynamic_vit.py:828,852): int(196*0.05)=9, int(196*0.03)=5.
MODEL_CASES = {
    # G1: micro geometry that keeps dh = 64
    "micro1": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=4, heads=2, num_classes=10,
                                  pruning_loc=(1,), token_ratio=(0.05,)), batch=3, seed=11),
    "micro2": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=4, heads=2, num_classes=10,
                                  pruning_loc=(1, 2), token_ratio=(0.05, 0.03)), batch=3, seed=12),
    "micro_small_pred": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=3, heads=2, num_classes=10,
                                            pruning_loc=(1,), token_ratio=(0.05,), small_predictor=True),
                             batch=2, seed=13),
    # --predictor-bn: the large predictor with BatchNormLayer instead of LayerNorm (dynamic_vit.py:438-476)
    "micro_bn": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=3, heads=2, num_classes=10,
                                    pruning_loc=(1,), token_ratio=(0.05,), predictor_bn=True), batch=4, seed=14),
    "micro_small_bn": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=3, heads=2, num_classes=10, pruning_loc=(1,), token_ratio=(0.05,),
                                          small_predictor=True, predictor_bn=True), batch=4, seed=15),
    # G2: BASELINE config 1 - DeiT-Tiny, 32x32, keep 1.0, batch 4 (N = 4, gather is the identity)
    "tiny32": dict(cfg=O.make_cfg(img_size=32, dim=192, depth=12, heads=3, pruning_loc=(3,),
                                  token_ratio=(1.0,)), batch=4, seed=21),
    # G3: DeiT-S 224
    "small_k70": dict(cfg=O.make_cfg(dim=384, depth=12, heads=6, pruning_loc=(3,), token_ratio=(0.7,)),
                      batch=2, seed=31),
    "small_k50": dict(cfg=O.make_cfg(dim=384, depth=12, heads=6, pruning_loc=(3,), token_ratio=(0.5,)),
                      batch=2, seed=32),
    "small_3stage": dict(cfg=O.make_cfg(dim=384, depth=12, heads=6, pruning_loc=(3, 6, 9),
                                        token_ratio=(0.7, 0.5, 0.3)), batch=2, seed=33),
    # BASELINE config 5 geometry: DeiT-Base 384x384 (577 tokens), keep 0.3 with the reference's hard-coded init_n = 196,
    # i.e. int(196 * 0.3) = 58 of 576 tokens survive (dynamic_vit.py:828,852)
    "base384_k30": dict(cfg=O.make_cfg(img_size=384, dim=768, depth=12, heads=12, pruning_loc=(3,), token_ratio=(0.3,)),
                        batch=1, seed=51),
}

# Cases with no reference fixture: the reference cannot run them (its keep count is hard-coded to int(196 * ratio)), the oracle can
# (make_cfg(init_n=...)).  BASELINE config 5 as bench.py times it by default: DeiT-Base 384x384, k = int(576 * 0.3) = 172.
ORACLE_CASES = {
    "base384_k30_n576": dict(cfg=O.make_cfg(img_size=384, dim=768, depth=12, heads=12, pruning_loc=(3,), token_ratio=(0.3,), init_n=576),
                             batch=1, seed=53),
}

# Dynamic keep ratio (--patch-score-threshold): name -> dict(cfg, batch, seed, threshold).  The keep probabilities of a stage sum to 1
# over the N tokens, so the threshold is the probability mass of the lowest-scored tokens that gets dropped.
THRESHOLD_CASES = {
    "micro_thr1": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=4, heads=2, num_classes=10, pruning_loc=(1,), token_ratio=(0.5,)),
                       batch=3, seed=61, threshold=0.3),
    "micro_thr2": dict(cfg=O.make_cfg(img_size=64, dim=128, depth=4, heads=2, num_classes=10, pruning_loc=(1, 2), token_ratio=(0.5, 0.3)),
                       batch=3, seed=62, threshold=0.45),
    "small_thr": dict(cfg=O.make_cfg(dim=384, depth=12, heads=6, pruning_loc=(3,), token_ratio=(0.5,)), batch=2, seed=63, threshold=0.35),
}

# tag -> (b, nS, d, k, sigma)
PTK_CASES = {
    "small": (2, 16, 12, 4, 0.05),
    "deit": (2, 16, 196, 98, 0.05),
    "wide": (1, 8, 576, 172, 0.02),
}


def make_weights(case, affine=True):
    """Student / teacher state dicts as numpy (reference key names).  Predictor weights use a larger std so
    that the k / k+1 score margins are far above fp32 accumulation-order noise."""
    cfg, seed = case["cfg"], case["seed"]
    sd_s = synth.fill_state_dict(O.student_param_shapes(cfg), seed=seed, std=0.02,
                                 std_overrides={"score_predictor": 0.08})
    sd_t = synth.fill_state_dict(O.teacher_param_shapes(cfg), seed=seed + 1000, std=0.02)
    if affine:
        sd_s = synth.perturb_affine(sd_s, seed=seed)
        sd_t = synth.perturb_affine(sd_t, seed=seed + 1000)
    return sd_s, sd_t


def make_images(case):
    cfg = case["cfg"]
    return synth.images(case["batch"], 3, cfg["img_size"], seed=case["seed"])


def make_labels(case):
    return synth.labels(case["batch"], case["cfg"]["num_classes"], seed=case["seed"])


def make_selection_probs(N, rows=64):
    """Score rows for the selection fixtures: softmax of random logits (so that fp32 softmax collapses some
    neighbours), rows with deliberate exact ties, a constant row and a two-level row."""
    logits = synth.normal(f"sel/{N}", (rows, N), std=0.3, seed=5).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True)).astype(np.float32)
    p = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
    # rows 0..7: copy values to create exact ties across the k boundary
    for r in range(min(8, rows)):
        src = (np.arange(N) * 7 + r) % N
        half = N // 2
        p[r, src[:half]] = p[r, src[half:2 * half]]
    if rows > 9:
        p[8, :] = np.float32(1.0 / N)
        p[9, :] = np.where(np.arange(N) % 3 == 0, np.float32(2.0 / N), np.float32(0.5 / N))
    return p


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


# T2T micro geometry: 64x64 image -> 256 -> 64 -> 16 tokens, D = 128, 2 heads, mlp_ratio 3
T2T_CASE = dict(img_size=64, dim=128, depth=2, heads=2, mlp_ratio=3.0, num_classes=10, batch=2, seed=41)


# BASELINE config 4 at its own geometry: T2T-ViT-14 (t2t_vit.py:182-199: embed_dim 384, depth 14, 6 heads, mlp_ratio 3), 224x224 images:
# 3136 -> 784 -> 196 tokens through the two token encoders (performer: T2t_vit_14, transformer: T2t_vit_t_14), 1000 classes
T2T_224_CASE = dict(img_size=224, dim=384, depth=14, heads=6, mlp_ratio=3.0, num_classes=1000, batch=1, seed=43)


def make_t2t_weights(tokens_type, pruning_loc=(), case=None):
    c = case or T2T_CASE
    shapes = O.t2t_param_shapes(c["img_size"], c["dim"], c["depth"], c["heads"], c["mlp_ratio"], c["num_classes"], tokens_type,
                                pruning_loc=pruning_loc)
    sd = synth.fill_state_dict(shapes, seed=c["seed"], std=0.02, std_overrides={"score_predictor": 0.08, "tokens_to_token": 0.05})
    sd = synth.perturb_affine(sd, seed=c["seed"])
    sd["pos_embed"] = O.sinusoid_encoding((c["img_size"] // 16) ** 2 + 1, c["dim"]).numpy()
    for k in list(sd):
        if k.endswith(".w"):   # the performer's frozen random features (token_performer.py:28-29): scale sqrt(m) * orthogonal-ish
            sd[k] = (synth.normal(k, sd[k].shape, std=1.0, seed=c["seed"]) * 0.7).astype(np.float32)
    return sd


def make_t2t_images(case=None, batch=None):
    c = case or T2T_CASE
    return synth.images(batch or c["batch"], 3, c["img_size"], seed=c["seed"])
