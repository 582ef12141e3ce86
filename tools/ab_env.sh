#!/bin/bash
# like tools/ab_libs.sh, with extra environment for every run: tools/ab_env.sh <out-name> "VAR=val ..." lib.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; ENVS=$2; shift 2
: > "$OUT"
cp "$R/path-tracing_amd/libpt_hip.so" /tmp/libpt_orig.so
for lib in /tmp/libpt_orig.so "$@"; do
  cp "$lib" "$R/path-tracing_amd/libpt_hip.so"
  for rep in 1 2; do
    env $ENVS timeout -k 10 200 python3 "$R/bench.py" --no-cpu-baseline --no-count --no-exact --steps 5 --warmup 1 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed" >> "$OUT"; tail -3 /tmp/ab.err >> "$OUT"; exit 1; }
    python3 -c "
import json,sys; d=json.load(open('/tmp/ab.json')); print('%-28s %8.1f Msamples/s  path %.3f ms/frame' % (sys.argv[1].split('/')[-1], d['value'], d['kernel_ms_per_frame']['path_ms']))" "$lib" >> "$OUT"
  done
done
cp /tmp/libpt_orig.so "$R/path-tracing_amd/libpt_hip.so"
cat "$OUT"
