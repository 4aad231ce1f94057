#!/bin/bash
# Two separate --pmc passes (kernel-trace only, as the pool requires) over a short bench run -> gpurun_out/<tag>_pmc_traffic<suffix>.json
# usage: tools/pmc_traffic.sh <tag> [suffix] [extra bench.py arguments, e.g. --config c5]
TAG=${1:-pmc}
SUF=${2:-}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/${TAG}${SUF}_$c -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $ROOT/gpurun_out/${TAG}${SUF}_$c.log 2>&1 || { tail -5 $ROOT/gpurun_out/${TAG}${SUF}_$c.log; exit 1; }
done
python3 $ROOT/tools/pmc_traffic.py $ROOT/gpurun_out/${TAG}${SUF}_FETCH_SIZE $ROOT/gpurun_out/${TAG}${SUF}_WRITE_SIZE $ROOT/gpurun_out/${TAG}_pmc_traffic${SUF}.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over 'bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing $*'; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md)"
find $ROOT/gpurun_out/${TAG}${SUF}_FETCH_SIZE $ROOT/gpurun_out/${TAG}${SUF}_WRITE_SIZE -name "*.csv" -size +2M -delete
