#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for i in 1 2; do
D2S_TEACHER_STREAM=0 python bench.py --no-cpu-baseline --steps 20 --no-kernel-timing > gpurun_out/x_b_ts0_$i.json 2>/dev/null
D2S_TEACHER_STREAM=1 python bench.py --no-cpu-baseline --steps 20 --no-kernel-timing > gpurun_out/x_b_ts1_$i.json 2>/dev/null
done
python -c "
import json
for f in ('ts0_1','ts1_1','ts0_2','ts1_2'):
    d=json.load(open('gpurun_out/x_b_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['config']['final_loss'])"
