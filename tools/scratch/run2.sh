#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
true
python tools/scratch/dbg_adam2.py > gpurun_out/x_dbg_adam2.log 2>&1; tail -50 gpurun_out/x_dbg_adam2.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "threshold or bf16 or t2t or perturbed or normal_noise" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/x_tests.log
