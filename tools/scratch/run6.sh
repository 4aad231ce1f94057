#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
D2S_GEMM_RK=0 python bench.py --no-cpu-baseline --steps 10 --gemm-shapes-out gpurun_out/x_shapes_rk0.txt > gpurun_out/x_b_rk0.json 2>/dev/null
D2S_GEMM_RK=1 python bench.py --no-cpu-baseline --steps 10 --gemm-shapes-out gpurun_out/x_shapes_rk1.txt > gpurun_out/x_b_rk1.json 2>/dev/null
paste gpurun_out/x_shapes_rk0.txt gpurun_out/x_shapes_rk1.txt | cut -c1-140
python -c "
import json
for f in ('rk0','rk1'):
    d=json.load(open('gpurun_out/x_b_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['all_gemm_layouts'])"
