#!/bin/bash
# Runs on the GPU box: counter passes for the vector-memory path (texture addresser TA, texture data TD,
# vector L1 TCP) over the gather microbenchmark (where that path is the only thing busy) and over one
# bench step, so that the path kernel's share of the gather-instruction rate can be read off directly.
# usage: tools/pmc_ta.sh <out-subdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P3="TA_BUSY_avr TA_BUFFER_COALESCED_READ_CYCLES_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE"
pass() { # name, counters, program...
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
pass gather_p1 "$P1" "$R/tools/ubench/bin/gather_rate" pmc || exit 1
pass gather_p2 "$P2" "$R/tools/ubench/bin/gather_rate" pmc || exit 1
pass gather_p3 "$P3" "$R/tools/ubench/bin/gather_rate" pmc || exit 1
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count --no-exact"
pass bench_p1 "$P1" python3 "$R/bench.py" $BARGS || exit 1
pass bench_p2 "$P2" python3 "$R/bench.py" $BARGS || exit 1
pass bench_p3 "$P3" python3 "$R/bench.py" $BARGS || exit 1
python3 "$R/tools/summarize_prof.py" "$OUT" pmc_all > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
