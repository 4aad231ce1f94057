#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_ddp.py -m gpu -q > gpurun_out/r03c_graph_tests.log 2>&1
echo "graph+ddp tests rc=$?"; tail -8 gpurun_out/r03c_graph_tests.log | cut -c1-300
for cfg in c3 c4 headline c2 c5; do
for g in off on; do
  timeout -k 10 300 python bench.py --config $cfg --graph $g --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03c_${cfg}_graph_$g.json 2> gpurun_out/r03c_${cfg}_graph_$g.err || { tail -20 gpurun_out/r03c_${cfg}_graph_$g.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03c_${cfg}_graph_$g.json"))
    print("$cfg graph=$g", d["value"], "img/s", d["ms_per_step"], "ms host", d.get("host_enqueue_ms_per_step"), d.get("step_issue","")[:20], "loss", d["config"]["final_loss"])
except Exception as e:
    print("$cfg graph=$g failed", e)
PY
done
done
