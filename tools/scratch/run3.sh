#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_f32.hip -o /tmp/mfma_shape_f32 2>/dev/null && /tmp/mfma_shape_f32 > gpurun_out/x_mfma_shape3.log 2>&1; grep "32x32" gpurun_out/x_mfma_shape3.log
for cap in 0 1 2 3 4; do D2S_GEMM_WG_PER_CU=$cap python tools/gemm_bench.py 0 > gpurun_out/x_gemm_cap$cap.log 2>&1; done
paste <(cut -c1-46 gpurun_out/x_gemm_cap0.log) <(cut -c27-46 gpurun_out/x_gemm_cap1.log) <(cut -c27-46 gpurun_out/x_gemm_cap2.log) <(cut -c27-46 gpurun_out/x_gemm_cap3.log) <(cut -c27-46 gpurun_out/x_gemm_cap4.log) <(cut -c47-80 gpurun_out/x_gemm_cap0.log) | grep -v amdgpu.ids
python tools/scratch/dbg_adam2.py > gpurun_out/x_dbg_adam2.log 2>&1; tail -50 gpurun_out/x_dbg_adam2.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "threshold or bf16 or t2t or perturbed or normal_noise" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/x_tests.log
