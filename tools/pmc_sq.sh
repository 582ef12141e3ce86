#!/bin/bash
# SQ time-breakdown counters for the path kernel; usage: tools/pmc_sq.sh <outdir-under-gpurun_out> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count --no-exact $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$name.log" 2>&1
  local rc=$?; echo "$name rc=$rc"; return $rc; }
run pmc_s1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE || exit 1
run pmc_s2 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS || exit 1
run pmc_s3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES || exit 1
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "^--|path_kernel" "$OUT/summary.txt"
