#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in serial overlap; do
  extra=""; [ $mode = serial ] && extra="--serial"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03d_prof_$mode -o p -- python3 $ROOT/bench.py --config c3 --graph off --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $ROOT/gpurun_out/r03d_prof_$mode.log 2>&1
  echo "rocprof $mode rc=$?"
  find $ROOT/gpurun_out/r03d_prof_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03d_c3_kernel_stats_$mode.csv
  find $ROOT/gpurun_out/r03d_prof_$mode -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03d_c3_kernel_trace_$mode.csv
  rm -rf $ROOT/gpurun_out/r03d_prof_$mode
  tail -2 $ROOT/gpurun_out/r03d_prof_$mode.log | cut -c1-400
done
ls -la $ROOT/gpurun_out/r03d_*
