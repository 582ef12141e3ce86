#!/bin/bash
# extra PMC passes for memory-pipeline diagnosis; usage: tools/pmc_extra.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count --no-exact $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$name.log" 2>&1
  local rc=$?; echo "$name rc=$rc"; return $rc; }
run pmc_x1 --pmc GRBM_TA_BUSY GRBM_TC_BUSY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY || exit 1
run pmc_x2 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || exit 1
run pmc_x3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum || exit 1
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "^--|path_kernel" "$OUT/summary.txt"
