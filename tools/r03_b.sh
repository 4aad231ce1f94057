#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python tools/graph_probe.py 2>&1 | tee gpurun_out/r03b_graph_probe.log
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q > gpurun_out/r03b_graph_tests.log 2>&1
echo "graph tests rc=$?"; tail -15 gpurun_out/r03b_graph_tests.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_ddp.py -m gpu -q -k "config3_batch32 or two_ranks" > gpurun_out/r03b_new_tests.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r03b_new_tests.log | cut -c1-300
