#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
D2S_GEMM_RK=0 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rk0.log 2>&1
D2S_GEMM_RK=1 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rk1.log 2>&1
D2S_GEMM_RK=0 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rk0b.log 2>&1
D2S_GEMM_RK=1 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_rk1b.log 2>&1
paste <(cut -c1-46 gpurun_out/x_gemm_rk0.log) <(cut -c27-46 gpurun_out/x_gemm_rk1.log) <(cut -c27-46 gpurun_out/x_gemm_rk0b.log) <(cut -c27-46 gpurun_out/x_gemm_rk1b.log) <(cut -c47-80 gpurun_out/x_gemm_rk0.log) | grep -v amdgpu.ids
D2S_GEMM_RK=1 timeout -k 10 900 python -m pytest tests -m gpu -q -k "gemm or linear or train_step_parity or optimizer" > gpurun_out/x_tests_rk.log 2>&1; echo "rk tests rc=$?"; tail -8 gpurun_out/x_tests_rk.log
timeout -k 10 900 python -m pytest tests -m gpu -q -k "threshold or bf16 or t2t or perturbed or normal_noise or transition" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/x_tests.log
