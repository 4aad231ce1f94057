#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
bash tools/pmc_gemm.sh 0 "NT teacher/student n=197 fc1" rkfc1 > gpurun_out/x_pmc1.log 2>&1
bash tools/pmc_gemm.sh 0 "NT teacher/student n=197 fc2" rkfc2 > gpurun_out/x_pmc2.log 2>&1
python tools/pmc_table.py gpurun_out/pmc_rkfc1_*/p_counter_collection.csv 2>&1 | tail -30
python tools/pmc_table.py gpurun_out/pmc_rkfc2_*/p_counter_collection.csv 2>&1 | tail -30
