#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
D2S_LN_PAIR=0 python tools/ln_bench.py 2>/dev/null | grep "D   384"
D2S_LN_PAIR=1 python tools/ln_bench.py 2>/dev/null | grep "D   384"
D2S_LN_PAIR=0 python tools/ln_bench.py 2>/dev/null | grep "D   384"
D2S_LN_PAIR=1 python tools/ln_bench.py 2>/dev/null | grep "D   384"
timeout -k 10 600 python -m pytest tests -m gpu -q -k "layernorm or train_step_parity" > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/x_tests.log
