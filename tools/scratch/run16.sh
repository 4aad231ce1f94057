#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for i in 1 2; do
D2S_ATTN_BWD_STREAMS=0 python bench.py --no-cpu-baseline --steps 20 --no-kernel-timing > gpurun_out/x_b_ab0_$i.json 2>/dev/null
D2S_ATTN_BWD_STREAMS=1 python bench.py --no-cpu-baseline --steps 20 --no-kernel-timing > gpurun_out/x_b_ab1_$i.json 2>/dev/null
done
python -c "
import json
for f in ('ab0_1','ab1_1','ab0_2','ab1_2'):
    d=json.load(open('gpurun_out/x_b_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['config']['final_loss'])"
timeout -k 10 900 python -m pytest tests -m gpu -q -k "optimizer or callers or ddp or overfit or full_size or t2t_14_full" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/x_tests.log
