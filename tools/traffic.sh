#!/bin/bash
# Runs on the GPU box.  (1) calibrates FETCH_SIZE on known byte counts in the path kernel's access
# patterns (tools/ubench/fetch_calib), (2) collects FETCH_SIZE and WRITE_SIZE of one bench run in
# separate passes, (3) writes <out>/traffic.json.
# usage: tools/traffic.sh <out-subdir-under-gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2 3; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib$mode" -o p -- "$R/tools/ubench/fetch_calib" $mode 4096 16 > "$OUT/calib$mode.log" 2>&1
  rc=$?; echo "calib$mode rc=$rc"; [ $rc -eq 0 ] || exit 1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/pmc_$c.log" 2>&1
  rc=$?; echo "pmc_$c rc=$rc"; [ $rc -eq 0 ] || exit 1
done
python3 "$R/tools/traffic.py" "$OUT" && cat "$OUT/traffic.json"
