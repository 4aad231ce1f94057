#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/x_b_a.json 2>/dev/null
D2S_DGRAD_NT_MIN_ROWS=1000000000 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/x_b_b.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/x_b_c.json 2>/dev/null
python -c "
import json
for f in ('a','b','c'):
    d=json.load(open('gpurun_out/x_b_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['all_gemm_layouts'])"
timeout -k 10 900 python -m pytest tests -m gpu -q -k "train_step_parity or optimizer or callers or ddp or overfit or full_size" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/x_tests.log
