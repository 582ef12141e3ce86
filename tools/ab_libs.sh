#!/bin/bash
# Runs on the GPU box: times bench.py with each library variant given (paths to .so files), by
# copying it over path-tracing_amd/libpt_hip.so in the box's scratch copy of the tree.
# usage: tools/ab_libs.sh <out-name> lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
: > "$OUT"
cp "$R/path-tracing_amd/libpt_hip.so" /tmp/libpt_orig.so
for lib in /tmp/libpt_orig.so "$@"; do
  cp "$lib" "$R/path-tracing_amd/libpt_hip.so"
  for rep in 1 2; do
    timeout -k 10 200 python3 "$R/bench.py" --no-cpu-baseline --no-count --no-exact --steps 5 --warmup 1 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed" >> "$OUT"; tail -3 /tmp/ab.err >> "$OUT"; exit 1; }
    python3 -c "
import json,sys; d=json.load(open('/tmp/ab.json')); print('%-28s %8.1f Msamples/s  path %.3f ms/frame' % (sys.argv[1].split('/')[-1], d['value'], d['kernel_ms_per_frame']['path_ms']))" "$lib" >> "$OUT"
  done
done
cp /tmp/libpt_orig.so "$R/path-tracing_amd/libpt_hip.so"
cat "$OUT"
