#!/bin/bash
# round 3, call A: new tests first (graph mode, k=172 parity, full-size C3/C5, DDP collectives), then config-3 graph vs eager A/B.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q > gpurun_out/r03a_graph_tests.log 2>&1
echo "graph tests rc=$?"; tail -15 gpurun_out/r03a_graph_tests.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_ddp.py tests/test_callers_gpu.py -m gpu -q -k "k172 or full_size or two_ranks or shorter_last" > gpurun_out/r03a_new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r03a_new_tests.log
for g in off on; do
  timeout -k 10 300 python bench.py --config c3 --graph $g --steps 30 --warmup 5 > gpurun_out/r03a_c3_graph_$g.json 2> gpurun_out/r03a_c3_graph_$g.err || { tail -20 gpurun_out/r03a_c3_graph_$g.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03a_c3_graph_$g.json"))
    print("c3 graph=$g", d["value"], "img/s", d["ms_per_step"], "ms host", d.get("host_enqueue_ms_per_step"), d.get("step_issue","")[:20], "loss", d["config"]["final_loss"])
except Exception as e:
    print("c3 graph=$g failed", e)
PY
done
for g in off on; do
  timeout -k 10 300 python bench.py --graph $g --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03a_head_graph_$g.json 2> gpurun_out/r03a_head_graph_$g.err || { tail -20 gpurun_out/r03a_head_graph_$g.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03a_head_graph_$g.json"))
    print("headline graph=$g", d["value"], "img/s", d["ms_per_step"], "ms host", d.get("host_enqueue_ms_per_step"), "loss", d["config"]["final_loss"])
except Exception as e:
    print("headline graph=$g failed", e)
PY
done
