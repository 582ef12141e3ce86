#!/bin/bash
# GPU box: bench + lane diagnostics for library variants under an environment.  usage: tools/gpu_try2.sh <out> "<env>" lib.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/$1; ENVS=$2; shift 2
: > "$OUT"
cp path-tracing_amd/libpt_hip.so /tmp/libpt_orig.so
for lib in /tmp/libpt_orig.so "$@"; do
  cp "$lib" path-tracing_amd/libpt_hip.so
  echo "== $lib ($ENVS)" >> "$OUT"
  env $ENVS timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-count --steps 4 --warmup 1 > /tmp/b.json 2> /tmp/b.err || { echo "bench failed" >> "$OUT"; tail -5 /tmp/b.err >> "$OUT"; cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so; cat "$OUT"; exit 1; }
  python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('%8.1f Msamples/s  path %.3f ms/frame' % (d['value'], d['kernel_ms_per_frame']['path_ms']))" >> "$OUT"
  env $ENVS FRAMES=4 timeout -k 10 200 python3 tools/kernel_stats.py 4 2>/dev/null | tail -8 >> "$OUT" || { echo "stats failed" >> "$OUT"; cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so; cat "$OUT"; exit 1; }
done
cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so
cat "$OUT"
