"""CPU tier: the drop-in boundary.  The C-ABI library loads and exports every symbol include/d2s_hip.h declares; the
product modules expose the reference's names / constructor signatures / state-dict keys; and the product path refuses to
run without a GPU instead of silently falling back."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from tests import cases
from oracle import d2s_oracle as O

REPO = cases.REPO


def test_library_exports_every_declared_symbol():
    from d2s import lib
    header = open(os.path.join(REPO, "include", "d2s_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(d2s_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 30
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/d2s_hip.h but not exported by libd2s_hip.so"
    assert sorted(lib.exported_symbols()) == declared, "python binding table and header disagree"


def test_state_dict_keys_match_reference():
    """Keys/shapes of the drop-in modules == the oracle's table == the parameter names the reference's own classes
    produced (recorded in the golden fixture's grad_names)."""
    import vit_models
    case = cases.MODEL_CASES["small_3stage"]
    cfg = case["cfg"]
    m = vit_models.VisionTransformerDiffPruning(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                                                pruning_loc=[3, 6, 9], token_ratio=[0.7, 0.5, 0.3], distill=True,
                                                topk_selection=True, predictor_loss_type="kl_div")
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = {k: tuple(s) for k, s in O.student_param_shapes(cfg)}
    assert got == want
    ref_names = [str(s) for s in cases.load_golden("model_small_3stage")["grad_names"]]
    assert [n for n, _ in m.named_parameters()] == ref_names          # same names, same registration order
    assert sum(p.numel() for p in m.parameters()) == 28549675           # counted on the reference's class
    t = vit_models.VisionTransformerTeacher(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True)
    assert {k: tuple(v.shape) for k, v in t.state_dict().items()} == {k: tuple(s) for k, s in O.teacher_param_shapes(cfg)}


def test_constructor_signature_matches_reference():
    import vit_models
    sig = inspect.signature(vit_models.VisionTransformerDiffPruning.__init__)
    ref = ["img_size", "patch_size", "in_chans", "num_classes", "embed_dim", "depth", "num_heads", "mlp_ratio", "qkv_bias",
           "qk_scale", "representation_size", "drop_rate", "attn_drop_rate", "drop_path_rate", "hybrid_backbone", "norm_layer",
           "pruning_loc", "token_ratio", "distill", "attn_selection", "attn_selection_threshold", "topk_selection", "early_exit",
           "mean_heads", "random_drop", "small_predictor", "predictor_loss_type", "predictor_bn", "patch_score_threshold"]
    assert list(sig.parameters)[1:1 + len(ref)] == ref          # dynamic_vit.py:648-653
    for name in ["VisionTransformerTeacher", "PredictorLG", "Attention", "Block", "Mlp", "PatchEmbed", "BatchNormLayer",
                 "batch_index_select", "resize_pos_embed", "checkpoint_filter_fn", "PerturbedTopK", "PerturbedTopKFunction",
                 "dynamic_vit_tiny_patch16_224_student", "dynamic_vit_small_patch16_224_student",
                 "dynamic_vit_base_patch16_224_student", "dynamic_vit_tiny_patch16_224_teacher",
                 "dynamic_vit_small_patch16_224_teacher", "dynamic_vit_base_patch16_224_teacher"]:
        assert hasattr(vit_models, name), name


def test_no_cpu_fallback():
    import vit_models
    from d2s.lib import D2SError
    m = vit_models.VisionTransformerTeacher(img_size=32, embed_dim=128, depth=1, num_heads=2, num_classes=10)
    with pytest.raises(D2SError):
        m(torch.zeros(1, 3, 32, 32))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "dense2sparse-vit_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):                         # no exception: the smoke checker lives in __graft_entry__.py
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} references the oracle"


def test_execution_order_arena_layout():
    import vit_models
    from d2s.engine import execution_order
    m = vit_models.VisionTransformerDiffPruning(img_size=64, embed_dim=128, depth=4, num_heads=2, num_classes=10,
                                                pruning_loc=[1, 2], token_ratio=[0.05, 0.03], topk_selection=True,
                                                predictor_loss_type="kl_div")
    order, starts = execution_order(m)
    assert order[:2] == ["cls_token", "pos_embed"] and order[-1] == "head.bias"
    i_pred0 = order.index("score_predictor.0.in_conv.0.weight")
    assert order.index("blocks.0.mlp.fc2.bias") < i_pred0 < order.index("blocks.1.norm1.weight")
    assert starts[1] == i_pred0 and starts[0] == order.index("blocks.0.norm1.weight")


def test_cli_accepts_every_reference_flag_with_the_same_default():
    """utils.parse_args mirrors the reference's command line (tests/golden/cli_flags.json: flag names, defaults, types and
    actions extracted from the text of the reference's utils.py:182-317 by tools/gen_cli_fixture.py)."""
    import json
    import utils
    spec = json.load(open(os.path.join(REPO, "tests", "golden", "cli_flags.json")))["flags"]
    assert len(spec) >= 50
    ns = vars(utils.parse_args([]))
    for ent in spec:
        flag = ent["flags"][0]
        dest = ent.get("dest", flag.lstrip("-").replace("-", "_"))
        assert dest in ns, f"{flag} is not accepted"
        if "default" in ent and flag != "--imgnet-val-dir":       # the reference's default is its author's home directory
            assert ns[dest] == ent["default"], (flag, ns[dest], ent["default"])
        if ent.get("action") == "store_true":
            assert vars(utils.parse_args([flag]))[dest] is True
        elif ent.get("type") in ("int", "float") and "nargs" not in ent:
            v = vars(utils.parse_args([flag, "3"]))[dest]
            assert v == 3 and type(v).__name__ == ent["type"]
    a = utils.parse_args(["--pruning-locs", "3", "6", "9", "--keep-ratios", "0.7", "0.5", "0.3", "--no-repeated-aug"])
    assert a.pruning_locs == [3, 6, 9] and a.keep_ratios == [0.7, 0.5, 0.3] and a.repeated_aug is False


def test_cli_rejects_flags_outside_the_path():
    import mask_predictor
    import utils
    for extra in (["--patch-score-threshold", "0.9", "--pruning-locs", "3", "6", "--keep-ratios", "0.7", "0.5"], ["--mask-loss-type", "bce"], ["--use-dp"]):
        with pytest.raises(SystemExit, match="not on the accelerated path"):
            mask_predictor.check_supported(utils.parse_args(extra))
    a = utils.parse_args([])
    mask_predictor.check_supported(a)
    assert a.mixup == 0.0 and a.cutmix == 0.0
    mask_predictor.check_supported(utils.parse_args(["--patch-score-threshold", "0.3", "--pruning-locs", "3", "--keep-ratios", "0.5"]))   # one stage: on the path


def test_checkpoint_ingestion_matches_reference_fixture():
    """checkpoint_filter_fn + resize_pos_embed against the reference's own output on a synthetic DeiT-style checkpoint
    (tests/golden/checkpoint.npz, tools/gen_golden.py::gen_checkpoint_ingestion): {'model': ...} unwrapping, matrix -> conv patch
    projection, 4x4 -> 6x6 position grid."""
    import types
    import vit_models
    z = np.load(os.path.join(REPO, "tests", "golden", "checkpoint.npz"))
    sd_in = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")}
    want = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    D, P = 16, 4
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(proj=types.SimpleNamespace(weight=torch.zeros(D, 3, P, P))),
                                  pos_embed=torch.zeros(1, 37, D))
    got = vit_models.checkpoint_filter_fn({"model": sd_in}, model)
    assert set(got) == set(want)
    for k in want:
        assert tuple(got[k].shape) == want[k].shape, k
        np.testing.assert_allclose(got[k].numpy(), want[k], rtol=1e-6, atol=1e-7, err_msg=k)
    # and through the factory: a local file, loaded with weights_only=True
    import tempfile
    m = vit_models.dynamic_vit_tiny_patch16_224_teacher()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["pos_embed"] = torch.randn(1, 1 + 7 * 7, 192)           # a 112 x 112 checkpoint
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ckpt.pth")
        torch.save({"model": sd}, path)
        m2 = vit_models.dynamic_vit_tiny_patch16_224_teacher(checkpoint_path=path)
    assert m2.pos_embed.shape == (1, 197, 192)
    np.testing.assert_allclose(m2.pos_embed[:, :1].detach().numpy(), sd["pos_embed"][:, :1].numpy())


def test_early_exit_head_is_a_parameter_container_like_the_reference():
    import vit_models
    m = vit_models.VisionTransformerDiffPruning(img_size=64, embed_dim=128, depth=2, num_heads=2, num_classes=10, pruning_loc=[1],
                                                token_ratio=[0.5], distill=True, topk_selection=True, predictor_loss_type="kl_div",
                                                early_exit=True)
    keys = set(m.state_dict())
    assert {"early_exit_head.0.weight", "early_exit_head.0.bias", "early_exit_head.1.weight", "early_exit_head.1.bias"} <= keys
    assert m.state_dict()["early_exit_head.1.weight"].shape == (10, 128)


def test_param_groups_and_lr_schedule_match_reference_fixture():
    """SURVEY 8a row 17, now pinned: tests/golden/param_groups.json was produced by the reference's own get_param_groups /
    adjust_learning_rate (their two function definitions taken from utils.py by ast and run against the reference's student class,
    tools/gen_golden.py::gen_param_groups).  Checked here: the host mirrors in utils.py, and the fused optimiser's grouping and
    schedule in d2s.engine - group membership by name, group learning rates, the frozen set and the top-k sigma for every epoch."""
    import json
    import types
    import utils
    import vit_models
    from d2s import engine
    fx = json.load(open(os.path.join(REPO, "tests", "golden", "param_groups.json")))
    case = cases.MODEL_CASES["micro2"]
    cfg = case["cfg"]

    def build():
        return vit_models.VisionTransformerDiffPruning(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                                                       num_heads=cfg["heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True,
                                                       num_classes=cfg["num_classes"], pruning_loc=list(cfg["pruning_loc"]),
                                                       token_ratio=list(cfg["token_ratio"]), distill=True, topk_selection=True,
                                                       predictor_loss_type="kl_div")
    a = fx["args"]
    args = types.SimpleNamespace(weight_decay=a["weight_decay"], lr=a["lr"], min_lr=a["min_lr"], epochs=a["epochs"],
                                 warmup_steps=a["warmup_steps"], topk_selection=True, initial_sigma=a["initial_sigma"], early_exit=False)
    # ---- host mirrors (utils.py)
    m = build()
    groups = utils.get_param_groups(m, args)
    by_id = {id(p): n for n, p in m.named_parameters()}
    for g in groups:
        assert [by_id[id(p)] for p in g["params"]] == fx["groups"][g["name"]]["params"], g["name"]
        assert g["weight_decay"] == fx["groups"][g["name"]]["weight_decay"]
    for epoch, want in enumerate(fx["epochs"]):
        utils.adjust_learning_rate(groups, args, epoch, m, warming_up_step=args.warmup_steps)
        for g in groups:
            if g["params"]:
                assert abs(g["lr"] - want["lr"][g["name"]]) <= 1e-12, (epoch, g["name"])
        assert sorted(n for n, p in m.named_parameters() if not p.requires_grad) == want["frozen"], epoch
        assert abs(args.current_sigma - want["sigma"]) <= 1e-12
    # ---- fused optimiser (d2s.engine): grouping by name and the same schedule
    m = build()
    for n, p in m.named_parameters():
        g = engine._group_of(n, p)
        expect = [k for k, v in fx["groups"].items() if n in v["params"]]
        assert ([g] if g is not None else []) == expect, n

    class Stub:
        def set_lrs(self, predictor_lr, backbone_lr):
            self.lrs = (predictor_lr, backbone_lr)
    opt = Stub()
    for epoch, want in enumerate(fx["epochs"]):
        engine.adjust_learning_rate(opt, m, epoch, a["epochs"], a["lr"], a["min_lr"], a["warmup_steps"])
        assert abs(opt.lrs[0] - want["lr"]["predictor"]) <= 1e-12 and abs(opt.lrs[1] - want["lr"]["base_decay"]) <= 1e-12, epoch
        assert sorted(n for n, p in m.named_parameters() if not p.requires_grad) == want["frozen"], epoch


def test_bf16_data_path_host_logic():
    """Host side of the bf16 data path, no GPU needed.  (1) The one-entry registry that carries a gradient's bf16 copy from one block's
    backward to the previous block's: a hit needs the same address, element count and version counter, an entry is used at most once,
    and a rewritten gradient misses.  (2) The bf16 side channels are refused outside the bf16 arithmetic mode (assertion before any
    library call).  (3) No cached bf16 weight is ever offered for a CPU tensor or a free-standing trainable one."""
    from d2s import ops
    g = torch.zeros(4, 6)
    g16 = torch.zeros(4, 6, dtype=torch.bfloat16)
    ops.shadow_put(g, g16)
    assert ops.shadow_take(g.view(24)) is g16            # a view of the same storage (what autograd hands to the next Function)
    assert ops.shadow_take(g) is None                    # used at most once
    ops.shadow_put(g, g16)
    g.add_(1.0)                                          # the gradient was rewritten (an accumulation, a hook): version counter moved
    assert ops.shadow_take(g) is None
    ops.shadow_put(g, g16)
    assert ops.shadow_take(torch.zeros(4, 6)) is None    # another tensor
    assert ops.shadow_take(g) is None                    # ... and any lookup empties the registry
    ops.shadow_put(g, g16)
    assert ops.shadow_take(g[:2]) is None                # same address, other element count
    assert ops.get_gemm_mode() == ops.GEMM_EXACT and not ops.bf16_io()
    with pytest.raises(AssertionError):
        ops.gemm(ops.NT, torch.zeros(32, 32), 32, torch.zeros(32, 32), 32, torch.zeros(32, 32), 32, 32, 32, 32,
                 a16=torch.zeros(32, 32, dtype=torch.bfloat16))
    w = torch.nn.Parameter(torch.zeros(64, 64))
    assert ops.bf16_weight(w) is None and ops.bf16_weight(w.detach()) is None and ops.bf16_weight(w, transposed=True) is None


def test_keep_ratio_summary_handles_a_shorter_last_batch():
    """drop_last=False loaders (ddp_training.py:15-20) end an epoch on a shorter batch: the per-image keep ratios of the batches are
    concatenated (torch.stack would raise after the whole epoch's work); avg = mean of the per-batch means as in train.py:67-70,77-80."""
    import torch
    from utils import keep_ratio_summary, SyntheticLoader
    batches = [torch.tensor([0.5, 0.7, 0.9, 0.3]), torch.tensor([0.2, 0.4, 0.6, 0.8]), torch.tensor([0.1, 1.0])]
    mn, avg, mx = keep_ratio_summary(batches)
    assert abs(mn - 0.1) < 1e-7 and abs(mx - 1.0) < 1e-7
    assert abs(avg - (0.6 + 0.5 + 0.55) / 3) < 1e-6
    sizes = [x.shape[0] for x, _ in SyntheticLoader(3, 4, img_size=8, last_batch=2)]
    assert sizes == [4, 4, 2]
