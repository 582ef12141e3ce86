#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/pipeline_sweep.txt; : > $OUT
export PT_PIPELINE_MB=16384
for C in 2 3 4 1 5; do
  K=12; [ $C = 5 ] && K=2
  for B in 6 5 4 3; do
    echo "== cfg $C PT_BLOCKS_PER_CU=$B" >> $OUT
    PT_BLOCKS_PER_CU=$B timeout -k 10 300 python3 tools/pipeline_probe.py $C $K 2>&1 | grep -E "PT_PIPELINE|equal: False" | cut -c1-75 >> $OUT || { echo "failed" >> $OUT; cat $OUT; exit 1; }
  done
done
cat $OUT
