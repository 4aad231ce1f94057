#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q -k composite > gpurun_out/r03e_tests.log 2>&1
echo "composite tests rc=$?"; tail -12 gpurun_out/r03e_tests.log | cut -c1-300
for cfg in c3 c4 headline; do
for comp in 0 1; do
  D2S_BLOCK_COMPOSITE=$comp timeout -k 10 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03e_${cfg}_comp$comp.json 2> gpurun_out/r03e_${cfg}_comp$comp.err || { tail -20 gpurun_out/r03e_${cfg}_comp$comp.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03e_${cfg}_comp$comp.json"))
    print("$cfg composite=$comp", d["value"], "img/s", d["ms_per_step"], "ms host", d.get("host_enqueue_ms_per_step"), "loss", d["config"]["final_loss"])
except Exception as e:
    print("$cfg composite=$comp failed", e)
PY
done
done
