#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 10 --gemm-shapes-out gpurun_out/x_shapes_gd.txt > gpurun_out/x_b_gd.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/x_b_gd.json')); print(d['value'], d['ms_per_step'], d['roofline']['all_gemm_layouts'])"
head -12 gpurun_out/x_shapes_gd.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -k "train_step_parity or kernels or t2t or split or bf16" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/x_tests.log
