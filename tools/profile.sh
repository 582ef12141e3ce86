#!/bin/bash
# Runs on the GPU box: kernel-trace statistics plus PMC passes of one short bench run each.
# usage: tools/profile.sh <out-subdir-under-gpurun_out> [bench args...]
# Every rocprofv3 pass is its own process under its own time limit; a failing pass ends the script.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count --no-exact $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS || exit 1
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
run pmc_ta --pmc TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
CFG=4; prev=""; for a in "$@"; do [ "$prev" = "--config" ] && CFG=$a; prev=$a; done
timeout -k 10 200 python3 "$R/tools/kernel_stats.py" $CFG --json "$OUT/kstats.json" > "$OUT/kstats.txt" 2>&1 || { tail -5 "$OUT/kstats.txt"; exit 1; }
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
python3 "$R/tools/pmc_json.py" "$OUT" "$OUT/path_kernel_pmc.json" > "$OUT/pmc_json.log" 2>&1 || { cat "$OUT/pmc_json.log"; exit 1; }
