#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for t in 768 1024 1280 1536 2048 768; do
D2S_SPLITK_TARGET=$t python bench.py --no-cpu-baseline --steps 20 --no-kernel-timing > gpurun_out/x_b_sk$t.json 2>/dev/null
python -c "
import json,sys
d=json.load(open('gpurun_out/x_b_sk$t.json')); print('splitk target $t', d['value'], d['ms_per_step'])"
done
