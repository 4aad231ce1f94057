#!/usr/bin/env python3
"""Which parts of the train step survive hipGraph capture?  Each stage runs in a child process (a failed capture can take the process
down), smallest first.  usage: python tools/graph_probe.py [stage]"""
import os
import subprocess
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "dense2sparse-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

STAGES = ["teacher_fwd", "student_fwd_nograd", "fwd_bwd_serial", "fwd_bwd_teacher_stream", "fwd_bwd_wgrad_stream", "full_step"]


def run(stage):
    import torch
    from tests import cases
    from tests.test_model_gpu import build_models, make_args
    from d2s import ops
    from d2s.engine import TrainStep
    dev = torch.device("cuda:0")
    case = cases.MODEL_CASES["micro2"]
    student, teacher, _, _ = build_models(case, dev)
    ts = TrainStep(student, teacher, make_args(case["cfg"]), graph=False)
    x = torch.from_numpy(cases.make_images(case)).to(dev)
    y = torch.from_numpy(cases.make_labels(case)).to(dev)
    side = torch.cuda.Stream()
    if stage in ("teacher_fwd", "student_fwd_nograd", "fwd_bwd_serial"):
        ts._teacher_stream = None
        ops._WGRAD_ENABLED = False
    if stage == "fwd_bwd_teacher_stream":
        ops._WGRAD_ENABLED = False
    if stage == "fwd_bwd_wgrad_stream":
        ts._teacher_stream = None

    def body():
        if stage == "teacher_fwd":
            with torch.no_grad():
                return teacher(x)[0]
        if stage == "student_fwd_nograd":
            with torch.no_grad():
                return student(x)[0]
        loss, info = ts._forward_backward(x, y, accumulate=False)
        return loss.detach()

    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            out = body()
        torch.cuda.synchronize()
        ref = out.clone()
        g = torch.cuda.CUDAGraph()
        for attr in ("pred_logits", "cls_attns", "kept_token_indices", "dropped_token_indices"):
            setattr(student, attr, [])
        with torch.cuda.graph(g, stream=side):
            out = body()
        print(f"[{stage}] captured", flush=True)
        g.replay()
        torch.cuda.synchronize()
        print(f"[{stage}] replayed: equal to eager = {torch.equal(out, ref)}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True, timeout=300)
            tail = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("[") or "rror" in ln][-4:]
            print(f"== {st}: rc {r.returncode}\n   " + "\n   ".join(tail), flush=True)
