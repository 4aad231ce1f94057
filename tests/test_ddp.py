"""Multi-process tier.  CPU: the gradient reducer's bucket logic over gloo, world_size 2.  GPU: two ranks sharing the
one GPU of the test box (gloo on device tensors) against a single-process run on the concatenated batch."""
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


class _FakeArena:
    def __init__(self, n, rank):
        self.total = n
        self.params = torch.zeros(n)
        self.grads = torch.arange(n, dtype=torch.float32) * (rank + 1)


def _worker(rank, world, port, q, collective="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d2s.engine import GradReducer
    arena = _FakeArena(10000, rank)
    red = GradReducer(arena, bucket_mb=4096 * 4 / (1 << 20), collective=collective)      # 4096-element buckets
    assert red.collective == collective and red.bucket_elems == 4096
    launches = []
    orig = red._launch
    red._launch = lambda lo: (launches.append((lo, red._hi)), orig(lo))[1]
    for lo in (9000, 7000, 5000, 2500, 100):                     # backward walks the arena from its end to its start
        red.ready_from(lo)
    scale = red.finish()
    expect = torch.arange(10000, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(arena.grads, expect) and scale == 1.0 / world
    # buckets: contiguous, non-overlapping, cover [0, total), each >= bucket size except the final flush
    spans = [(lo, hi) for lo, hi in launches if hi > lo]
    ok = ok and spans[0][1] == 10000 and spans[-1][0] == 0 and all(spans[i][0] == spans[i + 1][1] for i in range(len(spans) - 1))
    ok = ok and all(hi - lo >= 4096 for lo, hi in spans[:-1])
    # second step reuses the reducer
    arena.grads = torch.ones(10000) * (rank + 1)
    red.ready_from(5000)
    red.finish()
    ok = ok and torch.equal(arena.grads, torch.ones(10000) * sum(r + 1 for r in range(world)))
    # warm-up epoch (utils.py:112-119): only the predictor slices hold live gradients and no autograd hook fires (nothing upstream
    # requires grad), so finish() must exchange exactly those slices and leave every frozen slice - stale values included - alone
    total = sum(r + 1 for r in range(world))
    red.set_live_ranges([(1000, 2000), (2000, 2500), (6000, 7000)])        # touching ranges merge
    assert red.live == [(1000, 2500), (6000, 7000)] and red.live_elems() == 2500
    arena.grads = torch.ones(10000) * (rank + 1)
    reduced = []
    orig_reduce = red._reduce
    red._reduce = lambda lo, hi: (reduced.append((lo, hi)), orig_reduce(lo, hi))[1]
    red.finish()
    want = torch.ones(10000) * (rank + 1)
    want[1000:2500] = total
    want[6000:7000] = total
    ok = ok and torch.equal(arena.grads, want) and reduced == [(6000, 7000), (1000, 2500)]
    # first full epoch: the live set grows (cls_token / pos_embed style gaps stay out), hooks fire again, buckets count live elements only
    red.set_live_ranges([(500, 9500)])
    arena.grads = torch.ones(10000) * (rank + 1)
    del reduced[:]
    for lo in (9000, 7000, 5000, 2500, 100):
        red.ready_from(lo)
    red.finish()
    want = torch.ones(10000) * (rank + 1)
    want[500:9500] = total
    ok = ok and torch.equal(arena.grads, want) and reduced[0] == (5000, 9500) and reduced[-1][0] == 500
    ok = ok and all(reduced[i][0] == reduced[i + 1][1] for i in range(len(reduced) - 1))
    # a slice the world size does not divide (odd length): rs_ag falls back to one all_reduce for it, same sums
    red.set_live_ranges([(0, 10000)])
    red.live = [(3, 1000)]
    arena.grads = torch.ones(10000) * (rank + 1)
    red.finish()
    want = torch.ones(10000) * (rank + 1)
    want[3:1000] = total
    ok = ok and torch.equal(arena.grads, want)
    ok = ok and red._n_collectives > 0
    q.put((rank, bool(ok), spans + reduced))
    dist.destroy_process_group()


@pytest.mark.parametrize("collective,port", [("allreduce", 29517), ("rs_ag", 29519)])
def test_grad_reducer_gloo_world2(collective, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, collective)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def test_grad_reducer_env_selects_bucket_and_collective(monkeypatch):
    from d2s.engine import GradReducer
    monkeypatch.setenv("D2S_DDP_BUCKET_MB", "4")
    monkeypatch.setenv("D2S_DDP_COLLECTIVE", "rs_ag")
    red = GradReducer(_FakeArena(64, 0))
    assert red.bucket_elems == (4 << 20) // 4 and red.collective == "rs_ag"
    monkeypatch.setenv("D2S_DDP_COLLECTIVE", "ring_of_fire")
    with pytest.raises(ValueError):
        GradReducer(_FakeArena(64, 0))


@pytest.mark.gpu
@pytest.mark.parametrize("collective,port", [("allreduce", 29541), ("rs_ag", 29543)])
def test_two_ranks_match_single_process_on_concatenated_batch(collective, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(cases.REPO, "tools", "ddp_check.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", D2S_DDP_COLLECTIVE=collective)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "[ddp_check]" in out.stdout


def _rccl_single_rank_worker(q, port):
    """Child process (an RCCL group must not leak into the pytest process): two steps of a data-parallel TrainStep on a one-rank RCCL group
    against the plain TrainStep on the same batches - same kernels, the collectives are identities, so everything is bit-identical -
    plus the stream placement (d2s.engine.TrainStep._place_beside_process_group) having run and reported."""
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from d2s import ops
        from d2s.engine import TrainStep
        from tests.test_model_gpu import build_models, make_args, _t
        from d2s import synth
        case = cases.MODEL_CASES["micro2"]
        cfg = case["cfg"]
        data = [(_t(synth.images(case["batch"], 3, cfg["img_size"], seed=700 + i)).to(dev), _t(synth.labels(case["batch"], cfg["num_classes"], seed=700 + i)).to(dev))
                for i in range(3)]
        plain = TrainStep(*build_models(case, dev)[:2], make_args(cfg), graph=False, warmup_steps=0)
        ddp = TrainStep(*build_models(case, dev)[:2], make_args(cfg), graph=False, warmup_steps=0, distributed=True)
        ddp.reducer.force = True
        same = True
        for x, y in data:
            a, b = plain(x, y), ddp(x, y)
            torch.cuda.synchronize()
            same = same and torch.equal(a["loss"], b["loss"]) and torch.equal(plain.arena.params, ddp.arena.params)
        probe = ddp.pg_probe
        picks = list(ops.stream_picks)
        q.put((same, probe, picks, ddp.reducer._n_collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_single_rank_rccl_step_is_bit_identical_and_places_its_streams():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(q, 29631))
    p.start()
    same, probe, picks, n_coll = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert same, "the data-parallel step on a one-rank RCCL group must reproduce the plain step bit for bit"
    assert n_coll > 0, "the reducer must have issued its collectives (force)"
    assert probe is not None and "caller_stream_shares_rccl_queue" in probe and "weight_grad_stream_shares_rccl_queue" in probe, probe
    assert any(p[0].startswith("teacher forward") for p in picks) and all(p[2] for p in picks), picks       # every pick verified on the device
