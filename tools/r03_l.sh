#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "attn or attention or bf16" > gpurun_out/r03l_tests.log 2>&1
echo "bf16 / attention tests rc=$?"; tail -6 gpurun_out/r03l_tests.log | cut -c1-300
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_attn_bench.txt
timeout -k 10 300 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03l_c5.json 2> gpurun_out/r03l_c5.err || tail -5 gpurun_out/r03l_c5.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03l_c5.json")); print("c5", d["value"], "img/s", d["ms_per_step"], "ms loss", d["config"]["final_loss"])
PY
