#!/bin/bash
# SQ counters for one GEMM shape in one arithmetic mode:  tools/pmc_gemm.sh <mode> "<shape filter>" <tag>
# (separate --pmc passes, kernel-trace only; run on the GPU box through gpurun)
set -e
MODE=$1; FILTER=$2; TAG=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export D2S_BENCH_ONLY="$FILTER"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$i -o p -- python3 $ROOT/tools/gemm_bench.py $MODE > $ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { tail -5 $ROOT/gpurun_out/pmc_${TAG}_$i.log; exit 1; }
done
ls $ROOT/gpurun_out/pmc_${TAG}_1
