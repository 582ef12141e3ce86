#!/bin/bash
# Runs on the GPU box: one short bench.py run per environment setting, one result line each.
# usage: tools/sweep_env.sh <out-name> [--config C] -- "VAR=val VAR2=val" "VAR=val" ...   ("-" = no extra environment)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
ARGS=""
while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
shift
: > "$OUT"
for ENVS in "$@"; do
  [ "$ENVS" = "-" ] && ENVS="PT_UNUSED=1"
  env $ENVS timeout -k 10 200 python3 "$R/bench.py" --no-cpu-baseline --no-count --no-exact --steps 4 --warmup 1 $ARGS > /tmp/sw.json 2> /tmp/sw.err || { echo "$ENVS failed" >> "$OUT"; tail -3 /tmp/sw.err >> "$OUT"; cat "$OUT"; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('/tmp/sw.json')); print('%-60s %8.1f Msamples/s  path %.3f ms/frame' % (sys.argv[1], d['value'], d['kernel_ms_per_frame']['path_ms']))" "$ENVS" >> "$OUT"
done
cat "$OUT"
