#!/bin/bash
# GPU box: parity subset, lane statistics of configurations 4 and 2, back-to-back timing of all configurations.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/quick.txt; : > $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_wgsl_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_batch_pipeline.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/quick.parity.log 2>&1 || { tail -40 gpurun_out/quick.parity.log; exit 1; }
tail -2 gpurun_out/quick.parity.log
for C in 4 2; do
  echo "== stats cfg $C" >> $OUT
  timeout -k 10 200 python3 tools/kernel_stats.py $C 2>&1 | grep -E "inner lanes|outer iters|wall clock" >> $OUT || { echo failed >> $OUT; cat $OUT; exit 1; }
done
for C in ${@:-1 2 3 4 5}; do
  K=12; [ $C = 5 ] && K=2
  echo "== cfg $C" >> $OUT
  timeout -k 10 300 python3 tools/pipeline_probe.py $C $K 2>&1 | grep -E "PT_PIPELINE|equal: False" | cut -c1-75 >> $OUT || { echo "failed" >> $OUT; cat $OUT; exit 1; }
done
cat $OUT
