#!/bin/bash
# GPU box: kernel statistics (counting build) and bench lines of a configuration for the library in the tree and for variants.
# usage: tools/gpu_ab_stats.sh <config> <frames> lib.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
C=$1; F=$2; shift 2
OUT=gpurun_out/ab_stats_cfg$C.txt; : > $OUT
cp path-tracing_amd/libpt_hip.so /tmp/libpt_orig.so
for lib in /tmp/libpt_orig.so "$@"; do
  cp "$lib" path-tracing_amd/libpt_hip.so
  echo "== $lib" >> $OUT
  FRAMES=$F timeout -k 10 300 python3 tools/kernel_stats.py $C 2>&1 | grep -E "trav share|inner lanes|outer iters|inner visits|wall clock" >> $OUT || { echo failed >> $OUT; break; }
  timeout -k 10 300 python3 bench.py --config $C --no-cpu-baseline --no-count --no-exact --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f Msamples/s  %.2f ms/step  path %.3f ms/frame' % (d['value'], d['ms_per_step'], d['kernel_ms_per_frame']['path_ms']))" >> $OUT || { echo failed >> $OUT; break; }
done
cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so
cat $OUT
