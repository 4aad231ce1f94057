#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
export D2S_BENCH_ONLY="teacher/student n=197 fc"
python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rand.log 2>&1
D2S_BENCH_ZEROS=1 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_zero.log 2>&1
python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rand2.log 2>&1
paste <(cut -c1-46 gpurun_out/x_gemm_rand.log) <(cut -c27-46 gpurun_out/x_gemm_zero.log) <(cut -c27-46 gpurun_out/x_gemm_rand2.log) <(cut -c47-80 gpurun_out/x_gemm_rand.log) | grep -v amdgpu.ids
unset D2S_BENCH_ONLY
D2S_TN_SMALL_TILE=0 python bench.py --no-cpu-baseline --steps 10 --gemm-shapes-out gpurun_out/x_shapes_tn0.txt > gpurun_out/x_b_tn0.json 2>/dev/null
D2S_TN_SMALL_TILE=1 python bench.py --no-cpu-baseline --steps 10 --gemm-shapes-out gpurun_out/x_shapes_tn1.txt > gpurun_out/x_b_tn1.json 2>/dev/null
paste <(grep "^TN" gpurun_out/x_shapes_tn0.txt | sort) <(grep "^TN" gpurun_out/x_shapes_tn1.txt | sort) | cut -c1-140
python -c "
import json
for f in ('tn0','tn1'):
    d=json.load(open('gpurun_out/x_b_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['all_gemm_layouts'])"
python bench.py --config c3 --steps 10 > gpurun_out/x_b_c3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/x_b_c3.json')); print('c3', d['value'], d['ms_per_step'], d['roofline']['all_gemm_layouts'], d['c_abi_calls_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/x_c3prof -o p -- python3 $ROOT/bench.py --config c3 --steps 10 --warmup 3 --no-kernel-timing > $ROOT/gpurun_out/x_c3prof.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/x_c3prof/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f))); tot=sum(int(r['TotalDurationNs']) for r in rows)
print('c3 kernel time per step (13 steps) ms:', tot/1e6/13, ' kernels/step:', sum(int(r['Calls']) for r in rows)/13)
PY
