#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
for t in 0 1 2 3 4; do D2S_GEMM_TILE=$t python tools/gemm_bench.py 0 > gpurun_out/x_gemm_tile$t.log 2>&1; done
paste <(cut -c1-46 gpurun_out/x_gemm_tile0.log) <(cut -c27-46 gpurun_out/x_gemm_tile1.log) <(cut -c27-46 gpurun_out/x_gemm_tile2.log) <(cut -c27-46 gpurun_out/x_gemm_tile3.log) <(cut -c27-46 gpurun_out/x_gemm_tile4.log) <(cut -c47-80 gpurun_out/x_gemm_tile0.log) | grep -v amdgpu.ids
