#!/bin/bash
# GPU box: back-to-back renders (tools/pipeline_probe.py) for the library in the tree and for variants, under environments.
# usage: tools/gpu_ab_probe.sh "<configs>" "<env1>|<env2>|..." lib.so ...      ("-" = no extra environment)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
CFGS=$1; IFS='|' read -ra ENVS <<< "$2"; shift 2
OUT=gpurun_out/ab_probe.txt; : > $OUT
cp path-tracing_amd/libpt_hip.so /tmp/libpt_orig.so
for C in $CFGS; do
  K=12; [ $C = 5 ] && K=2
  for lib in /tmp/libpt_orig.so "$@"; do
    cp "$lib" path-tracing_amd/libpt_hip.so
    for E in "${ENVS[@]}"; do
      [ "$E" = "-" ] && E="PT_UNUSED=1"
      echo "== cfg $C $(basename $lib) $E" >> $OUT
      env $E timeout -k 10 300 python3 tools/pipeline_probe.py $C $K 2>&1 | grep -E "PT_PIPELINE|equal: False" | cut -c1-72 >> $OUT || { echo failed >> $OUT; cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so; cat $OUT; exit 1; }
    done
  done
done
cp /tmp/libpt_orig.so path-tracing_amd/libpt_hip.so
cat $OUT
