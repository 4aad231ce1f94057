#!/bin/bash
# scratch experiments on the GPU box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python tools/scratch/dbg_adam.py > gpurun_out/x_dbg_adam.log 2>&1; tail -8 gpurun_out/x_dbg_adam.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_f32.hip -o /tmp/mfma_shape_f32 && /tmp/mfma_shape_f32 > gpurun_out/x_mfma_shape.log 2>&1; cat gpurun_out/x_mfma_shape.log
D2S_GEMM_MFMA16=0 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_mf32.log 2>&1
D2S_GEMM_MFMA16=1 python tools/gemm_bench.py 0 > gpurun_out/x_gemm_mf16.log 2>&1
paste <(cut -c1-46 gpurun_out/x_gemm_mf32.log) <(cut -c20-46 gpurun_out/x_gemm_mf16.log) <(cut -c47-90 gpurun_out/x_gemm_mf32.log)
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "threshold or bf16 or t2t or perturbed or normal_noise or split or gemm" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/x_tests.log
