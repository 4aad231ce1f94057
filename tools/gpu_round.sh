#!/bin/bash
# One gpurun call: GPU test suite, default bench line, rocprofv3 kernel stats of the same command.  usage: tools/gpu_round.sh <tag> [pytest args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
cat gpurun_out/${TAG}_bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o p -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --serial > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
echo "rocprof rc=$?"
find $ROOT/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $ROOT/gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
head -25 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
