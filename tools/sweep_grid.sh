#!/bin/bash
# usage: tools/sweep_grid.sh "VAR1 VAR2 ..." "v1,v2 v1,v2 ..."  -> one bench run per value tuple
VARS=($1); shift
for tup in $1; do
  IFS=, read -ra vals <<< "$tup"
  envs=()
  for i in "${!VARS[@]}"; do envs+=("${VARS[$i]}=${vals[$i]}"); done
  out=$(env "${envs[@]}" timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count 2>/dev/null | tail -1)
  echo "${envs[*]} $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f Msamples/s  path %.3f" % (d["value"], d["kernel_ms_per_frame"]["path_ms"]))')"
done
