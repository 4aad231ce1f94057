#!/bin/bash
# End-of-round evidence in one gpurun call: GPU suite, default bench line, rocprofv3 kernel stats, PMC traffic passes, the other BASELINE
# configs and arithmetic modes on one GPU.  usage: tools/evidence_round.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT && python -c "import __graft_entry__ as g; g.smoke()" || exit 1
cd $ROOT && bash tools/gpu_round.sh $TAG || exit 1
bash tools/pmc_traffic.sh ${TAG} "" || exit 1
bash tools/pmc_traffic.sh ${TAG} _c5 --config c5 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_c5 -o p -- python3 $ROOT/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --serial > $ROOT/gpurun_out/${TAG}_prof_c5.log 2>&1
find $ROOT/gpurun_out/${TAG}_prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/${TAG}_kernel_stats_c5.csv
find $ROOT/gpurun_out/${TAG}_prof_c5 -name "*kernel_trace.csv" -delete
cd $ROOT
for c in c2 c3 c4 c5; do timeout -k 10 300 python bench.py --config $c --steps 8 --warmup 3 > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || tail -3 gpurun_out/${TAG}_bench_$c.err; done
timeout -k 10 300 python bench.py --config c5 --reference-quirk --steps 8 --warmup 3 > gpurun_out/${TAG}_bench_c5_quirk.json 2>/dev/null
timeout -k 10 300 python bench.py --gemm-mode split --no-cpu-baseline > gpurun_out/${TAG}_bench_split.json 2>/dev/null
timeout -k 10 300 python bench.py --gemm-mode bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2>/dev/null
TAG_=$TAG python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/%s_bench*.json" % os.environ.get("TAG_", ""))):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(os.path.basename(f), d["value"], d["ms_per_step"], r.get("kernel"), r.get("achieved"), r.get("peak"), r.get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
