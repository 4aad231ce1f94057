#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "bf16 or gemm_split or wgrad" > gpurun_out/p_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/p_tests.log
[ $rc -ne 0 ] && exit 1
for pe in 0 1; do
D2S_SPLIT_DMA_TN=$pe D2S_BENCH_ONLY="TN" D2S_BENCH_MODEL=base python tools/gemm_bench.py 2 > gpurun_out/p_gb_$pe.txt 2>&1
D2S_SPLIT_DMA_TN=$pe python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/p_b_c5_$pe.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/p_b_c5_$pe.json')); print('c5 dma_tn=$pe', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['all_gemm_layouts'])"
done
paste gpurun_out/p_gb_0.txt gpurun_out/p_gb_1.txt | cut -c1-200
