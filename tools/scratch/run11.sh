#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "bf16_gemm_mode or train_step_parity or eval_forward or bf16" > gpurun_out/x_tests_s.log 2>&1; echo rc=$?
grep -h "^\[" gpurun_out/x_tests_s.log | cut -c1-400
