#!/bin/bash
# Runs on the GPU box: the microbenchmarks whose figures the roofline of the path kernel is priced against.
# usage: tools/ubench/run_all.sh <out-subdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 "$R/tools/ubench/bin/valu_rate" > "$OUT/valu_rate.txt" 2>&1 || exit 1
timeout -k 10 240 "$R/tools/ubench/bin/gather_rate" > "$OUT/gather_rate.txt" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/valu_pmc" -o p -- "$R/tools/ubench/bin/valu_rate" > "$OUT/valu_pmc.log" 2>&1 || exit 1
cat "$OUT/valu_rate.txt" "$OUT/gather_rate.txt"
