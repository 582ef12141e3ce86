#!/bin/bash
# GPU box: fixed cost per launch and two-stream overlap for path-kernel builds with 4, 2 and 1 waves per block.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/blocksize.txt; : > $OUT
cp path-tracing_amd/libpt_hip.so /tmp/libpt_orig.so
for V in orig w2 w1; do
  [ $V = orig ] || cp path-tracing_amd/variants/libpt_$V.so path-tracing_amd/libpt_hip.so
  echo "== $V" >> $OUT
  timeout -k 10 200 python3 tools/fixed_cost.py 2 2>&1 | grep max_bounce >> $OUT || { echo failed >> $OUT; break; }
  timeout -k 10 200 python3 tools/fixed_cost.py 4 2>&1 | grep "max_bounce_count 8" >> $OUT || { echo failed >> $OUT; break; }
  timeout -k 10 200 python3 tools/overlap_probe.py 2 16 2>&1 | grep "one stream" >> $OUT || { echo failed >> $OUT; break; }
  timeout -k 10 200 python3 tools/overlap_probe.py 4 8 2>&1 | grep "one stream" >> $OUT || { echo failed >> $OUT; break; }
done
cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so
cat $OUT
