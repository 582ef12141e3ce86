#!/bin/bash
# GPU box: the same two counter passes over bench.py for the library in the tree and for one variant under an
# environment; per-kernel means end up in gpurun_out/<out>/summary_*.txt.
# usage: tools/pmc_compare.sh <out> <variant.so> "<env of the variant run>"
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; LIB=$(cd "$R" && realpath "$2"); VENV=$3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-count --no-exact"
passes() { # tag
  local tag=$1
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$tag/pmc_a" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$tag.a.log" 2>&1 || return 1
  timeout -s KILL 240 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$tag/pmc_b" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$tag.b.log" 2>&1 || return 1
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/trace" -o p -- python3 "$R/bench.py" $BARGS > "$OUT/$tag/trace.log" 2>&1 || return 1
  python3 "$R/tools/summarize_prof.py" "$OUT/$tag" > "$OUT/summary_$tag.txt" 2>&1
  grep -E "^--|path_kernel" "$OUT/summary_$tag.txt"
}
passes base || { echo "base passes failed"; exit 1; }
cp "$R/path-tracing_amd/libpt_hip.so" /tmp/libpt_orig.so
cp "$LIB" "$R/path-tracing_amd/libpt_hip.so"
export $VENV
passes variant; rc=$?
cp /tmp/libpt_orig.so "$R/path-tracing_amd/libpt_hip.so"
exit $rc
