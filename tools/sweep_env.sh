#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]   -> prints value, Msamples/s and kernel ms per frame
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  out=$(env $VAR=$v timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count "$@" 2>/dev/null | tail -1)
  echo "$VAR=$v $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f Msamples/s" % d["value"], {k: round(v,3) for k,v in d["kernel_ms_per_frame"].items()})')"
done
