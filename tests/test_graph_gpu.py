"""The train step captured into a hipGraph (d2s.engine.TrainStep graph mode) must be the SAME step as the eager one: identical
kernels with identical arguments, only issued by one replay instead of ~450 C-ABI calls - so losses, kept ids and every parameter
after every optimiser step are bit-identical, across fresh inputs per step, the warm-up -> full epoch switch (a different live
gradient set = a different captured graph), learning-rate changes (device-side descriptors updated in place) and the data-parallel
reducer (collectives issued after the replay)."""
import types

import numpy as np
import pytest
import torch

from tests import cases
from tests.test_model_gpu import build_models, make_args, _t

pytestmark = pytest.mark.gpu


def _pair(case, dev, **kw):
    from d2s.engine import TrainStep
    out = []
    for graph in (False, True):
        student, teacher, _, _ = build_models(case, dev)
        out.append(TrainStep(student, teacher, make_args(case["cfg"]), graph=graph, **kw))
    return out


def _batches(case, n, dev):
    from d2s import synth
    cfg = case["cfg"]
    return [(_t(synth.images(case["batch"], 3, cfg["img_size"], seed=900 + i)).to(dev),
             _t(synth.labels(case["batch"], cfg["num_classes"], seed=900 + i)).to(dev)) for i in range(n)]


def _same_step(eager, graph, x, y, tag):
    ie, ig = eager(x, y), graph(x, y)
    torch.cuda.synchronize()
    assert torch.equal(ie["loss"], ig["loss"]), (tag, float(ie["loss"]), float(ig["loss"]))
    assert torch.equal(ie["mask_loss"].detach(), ig["mask_loss"].detach()) and torch.equal(ie["backbone_loss"].detach(), ig["backbone_loss"].detach()), tag
    for a, b in zip(ie["kept"], ig["kept"]):
        assert torch.equal(a, b), tag
    assert torch.equal(ie["logits_s"].detach(), ig["logits_s"].detach()), tag
    if not torch.equal(eager.arena.grads, graph.arena.grads):
        d = (eager.arena.grads - graph.arena.grads).abs()
        bad = torch.nonzero(d > 0).flatten()
        names = sorted({eager.arena.names[max(i for i, o in enumerate(eager.arena.offsets) if o <= int(j))] for j in bad[:: max(1, bad.numel() // 64)]})
        raise AssertionError(f"{tag}: gradients differ in {bad.numel()} elements, max abs {float(d.max()):.3e} (max |g| {float(eager.arena.grads.abs().max()):.3e}); "
                             f"first offset {int(bad[0])}; tensors: {names[:12]}")
    assert torch.equal(eager.arena.params, graph.arena.params), f"{tag}: parameters differ after the update"
    assert torch.equal(eager.opt.exp_avg_sq, graph.opt.exp_avg_sq), tag


def test_graph_step_is_bit_identical_to_eager_across_epochs():
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro2"]
    eager, graph = _pair(case, dev, warmup_steps=1, lr=5e-4, min_lr=1e-5, epochs=6)
    data = _batches(case, 5, dev)
    captured = []
    for epoch in (0, 1, 2):                                  # 0: warm-up (predictors only), 1: everything (new graph), 2: new learning rates
        eager.set_epoch(epoch)
        graph.set_epoch(epoch)
        for i, (x, y) in enumerate(data):
            _same_step(eager, graph, x, y, f"epoch {epoch} step {i}")
            captured.append(graph.last_step_captured)
            assert not eager.last_step_captured
    warm = graph.GRAPH_WARM_STEPS
    assert captured[:5] == [False] * warm + [True] * (5 - warm), captured          # epoch 0: eager warm steps, then replays
    assert captured[5:10] == [False] * warm + [True] * (5 - warm), captured        # epoch 1: the live set changed -> captured again
    assert captured[10:] == [True] * 5, captured                                  # epoch 2: same graph, new learning rates in place
    assert len(graph._graphs) == 2
    # the host-side running means of the loss modules advance once per replay
    for k, v in eager.metrics.items():
        np.testing.assert_array_equal(float(v), float(graph.metrics[k]), err_msg=k)
    assert eager.mask_loss_fn.count == graph.mask_loss_fn.count == 16
    # per-tensor AdamW counters on the device: predictor tensors 15 updates, backbone tensors 10
    assert torch.equal(eager.opt.chunk_steps, graph.opt.chunk_steps) and int(graph.opt.chunk_steps.max()) == 15


def test_graph_step_handles_a_shorter_last_batch_and_fresh_loss_modules():
    """A batch of another shape (drop_last=False loaders) runs eagerly / gets its own graph; train_one_epoch swaps the loss modules and
    the metrics dict every epoch (train.py:14-15) - the captured kernels do not depend on them."""
    from losses import MaskLoss, BackboneLoss
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro1"]
    eager, graph = _pair(case, dev, warmup_steps=0)
    data = _batches(case, 4, dev)
    for i, (x, y) in enumerate(data):
        _same_step(eager, graph, x, y, f"step {i}")
    assert graph.last_step_captured
    xs, ys = data[0][0][:2].contiguous(), data[0][1][:2].contiguous()
    _same_step(eager, graph, xs, ys, "short batch")
    assert not graph.last_step_captured
    for ts in (eager, graph):
        args = make_args(case["cfg"])
        ts.metrics = {}
        ts.mask_loss_fn, ts.backbone_loss_fn = MaskLoss(args, "train"), BackboneLoss(args)
    _same_step(eager, graph, *data[1], "after the loss modules were replaced")
    assert graph.last_step_captured and graph.mask_loss_fn.count == 2
    for k, v in eager.metrics.items():
        np.testing.assert_array_equal(float(v), float(graph.metrics[k]), err_msg=k)


def test_graph_step_with_the_gradient_reducer():
    """Data-parallel mode: nothing is exchanged from inside the capture; after the replay the reducer flushes the live gradient set in
    one go (single-rank gloo group with the collectives forced: the sums are the identity, the code path is the N-rank one)."""
    import os
    import tempfile
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro2"]
    store = tempfile.NamedTemporaryFile(delete=False)
    dist.init_process_group("gloo", init_method=f"file://{store.name}", rank=0, world_size=1)
    try:
        eager, graph = _pair(case, dev, warmup_steps=1, distributed=True, bucket_mb=0.25)
        flushed = []
        for ts in (eager, graph):
            ts.reducer.force = True
        orig = graph.reducer._reduce
        graph.reducer._reduce = lambda lo, hi: (flushed.append((lo, hi)), orig(lo, hi))[1]
        data = _batches(case, 4, dev)
        for epoch in (0, 1):
            eager.set_epoch(epoch)
            graph.set_epoch(epoch)
            for i, (x, y) in enumerate(data):
                del flushed[:]
                _same_step(eager, graph, x, y, f"epoch {epoch} step {i}")
                if graph.last_step_captured:          # one flush covering exactly the live ranges, issued after the replay
                    assert sorted(flushed) == sorted(graph.reducer.live), (flushed, graph.reducer.live)
        assert graph.last_step_captured
    finally:
        dist.destroy_process_group()
        if os.path.exists(store.name):
            os.unlink(store.name)


def test_graph_step_full_size_config3():
    """BASELINE config 3 at its per-rank batch (DeiT-S 224, stages 0.7 / 0.5 / 0.3, 32 images): the regime the graph mode exists for.
    Bit-identical to the eager step, and `auto` picks the graph at this size."""
    import vit_models
    from d2s import synth
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    steps = []
    for graph in (False, "auto"):
        torch.manual_seed(0)
        student = vit_models.dynamic_vit_small_patch16_224_student([3, 6, 9], [0.7, 0.5, 0.3], topk_selection=True, predictor_loss_type="kl_div").to(dev)
        teacher = vit_models.dynamic_vit_small_patch16_224_teacher().to(dev)
        args = types.SimpleNamespace(keep_ratios=[0.7, 0.5, 0.3], mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
        steps.append(TrainStep(student, teacher, args, warmup_steps=0, graph=graph))
    eager, graph = steps
    assert torch.equal(eager.arena.params, graph.arena.params)
    for i in range(5):
        x = _t(synth.images(32, 3, 224, seed=40 + i)).to(dev)
        y = _t(synth.labels(32, 1000, seed=40 + i)).to(dev)
        _same_step(eager, graph, x, y, f"step {i}")
    assert graph.last_step_captured, "auto mode must capture the 32-images-per-GPU step"
    big = torch.empty((128, 3, 224, 224), device=dev)
    assert not graph._use_graph(big), "auto mode leaves the 128-images-per-GPU step eager"


@pytest.mark.parametrize("name,mode", [("micro2", 0), ("micro1", 1), ("small_3stage", 0)])
def test_composite_block_calls_are_bit_identical_to_the_per_op_sequence(name, mode):
    """csrc/block.hip (one C-ABI call per block forward / backward) issues the launches of the per-op path from C: same kernels, same
    arguments, same order - so losses, ids and every gradient are bit-identical (exact and bf16x3-split arithmetic; warm-up epoch =
    forward-only blocks, full epoch = everything; weight gradients on their own stream inside TrainStep)."""
    from d2s import ops
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES[name]
    x, y = _t(cases.make_images(case)).to(dev), _t(cases.make_labels(case)).to(dev)
    outs = []
    ops.set_gemm_mode(mode)
    try:
        for composite in (False, True):
            ops._BLOCK_COMPOSITE = composite
            student, teacher, _, _ = build_models(case, dev)
            ts = TrainStep(student, teacher, make_args(case["cfg"]), warmup_steps=1, graph=False)
            rec = []
            for epoch in (0, 1):
                ts.set_epoch(epoch)
                for _ in range(2):
                    info = ts(x, y)
                    torch.cuda.synchronize()
                    rec.append((info["loss"].clone(), [k.clone() for k in info["kept"]], ts.arena.grads.clone(), ts.arena.params.clone()))
            outs.append(rec)
    finally:
        ops._BLOCK_COMPOSITE = True
        ops.set_gemm_mode(ops.GEMM_EXACT)
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a[0], b[0]), (i, float(a[0]), float(b[0]))
        assert all(torch.equal(p, q) for p, q in zip(a[1], b[1])), i
        assert torch.equal(a[2], b[2]), f"step {i}: gradients differ"
        assert torch.equal(a[3], b[3]), f"step {i}: parameters differ"


def test_teacher_lookahead_is_bit_identical_and_survives_a_wrong_announcement():
    """TrainStep(images, labels, next_images): the frozen teacher's forward for the next batch is issued one step early, on the teacher
    stream, beside this step's backward.  Same kernels on the same inputs, so every step is bit-identical to the plain one - also when
    the batch that arrives is not the one that was announced (its early teacher pass is dropped and recomputed)."""
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro2"]
    plain, ahead = [TrainStep(*build_models(case, dev)[:2], make_args(case["cfg"]), graph=False, warmup_steps=0) for _ in range(2)]
    data = _batches(case, 5, dev)

    class Both:                                                    # _same_step calls (x, y); this one also announces the next batch
        def __init__(self, ts):
            self.ts, self.nxt = ts, None
        def __call__(self, x, y):
            return self.ts(x, y, self.nxt)
        def __getattr__(self, k):
            return getattr(self.ts, k)
    both = Both(ahead)
    used = []
    for i, (x, y) in enumerate(data):
        both.nxt = data[i + 1][0] if i + 1 < len(data) and i != 2 else (data[0][0] if i == 2 else None)     # step 2 announces the WRONG batch
        had = ahead._ahead is not None and ahead._ahead["key"] == ahead._batch_key(x)
        _same_step(plain, both, x, y, f"step {i}")
        used.append(had)
    assert used == [False, True, True, False, True], used
    assert ahead._ahead is None
