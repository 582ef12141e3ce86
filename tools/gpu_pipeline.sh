#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/pipeline.txt; : > $OUT
for C in 2 3 4 1; do
  echo "== cfg $C" >> $OUT
  timeout -k 10 200 python3 tools/pipeline_probe.py $C 16 >> $OUT 2>&1 || { echo "failed" >> $OUT; cat $OUT; exit 1; }
done
timeout -k 10 200 python3 tools/interactive_latency.py >> $OUT 2>&1 || { echo "failed" >> $OUT; cat $OUT; exit 1; }
cat $OUT
