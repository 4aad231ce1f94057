"""bench.py --gpus N must start N ranks itself when no launcher is around it (the reference spawns its workers the
same way: mask_predictor.py:160-162, ddp_training.py:4-8).  CPU rehearsal over gloo: --launch-check."""
import json
import os
import subprocess
import sys

from tests import cases

BENCH = os.path.join(cases.REPO, "bench.py")


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + extra, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_launches_two_ranks():
    out = _run(["--gpus", "2", "--launch-check"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == [0, 1] and rec["launch_check"] is True


def test_bench_refuses_world_size_mismatch():
    # a launcher that started a different number of ranks than --gpus says: no line, non-zero exit
    out = _run(["--gpus", "2", "--launch-check"], env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_PORT="29561"))
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_child_failure_propagates():
    # the ranks die (no GPU in the CPU container / on any box the real path needs one): the parent must exit non-zero and print no line
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
