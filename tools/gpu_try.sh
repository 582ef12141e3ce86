#!/bin/bash
# GPU box: parity subset under an environment, then a bench sweep.  usage: tools/gpu_try.sh "<env for parity>" <out-name> -- env1 env2 ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
PENV=$1; OUT=$2; shift 3
env $PENV timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_wgsl_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$OUT.parity.log 2>&1 || { tail -40 gpurun_out/$OUT.parity.log; exit 1; }
tail -2 gpurun_out/$OUT.parity.log
timeout -k 10 900 bash tools/sweep_env.sh $OUT.txt -- "$@"
