#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/policy.txt; : > $OUT
for C in 1 2 3 4 5; do
  K=12; [ $C = 5 ] && K=2
  for A in 1 0; do
    echo "== cfg $C PT_ADAPT_GRID=$A" >> $OUT
    PT_ADAPT_GRID=$A timeout -k 10 300 python3 tools/pipeline_probe.py $C $K 2>&1 | grep -E "PT_PIPELINE|equal: False" | cut -c1-75 >> $OUT || { echo "failed" >> $OUT; cat $OUT; exit 1; }
  done
done
timeout -k 10 200 python3 tools/interactive_latency.py >> $OUT 2>&1 || { echo "failed" >> $OUT; cat $OUT; exit 1; }
cat $OUT
