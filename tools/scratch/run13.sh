#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
D2S_FORCE_NCCL=1 NCCL_DEBUG=WARN timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 tools/ddp_check.py > gpurun_out/x_nccl2.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/x_nccl2.log | cut -c1-300
