#!/bin/bash
# Kernel trace of back-to-back renders (tools/pipeline_probe.py): when do the kernels of consecutive batches run?
# usage: tools/gpu_overlap_trace.sh <config> <renders> [tag]
set -e
REPO=$(realpath "$(dirname "$0")/..")
OUT=$REPO/gpurun_out/overlap${3:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 $REPO/tools/pipeline_probe.py ${1:-2} ${2:-8} > $OUT/probe.log 2>&1
grep PT_PIPELINE $OUT/probe.log
python3 - $OUT/trace_kernel_trace.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('path_kernel','resolve','primary','fill'))]
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-34:-14]:
    print(r['Queue_Id'], r['Kernel_Name'][:22].ljust(22), "%.3f %.3f  dur %.3f" % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
