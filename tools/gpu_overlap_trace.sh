#!/bin/bash
# Kernel trace of tools/overlap_probe.py: do the path kernels of two streams overlap in time?
set -e
REPO=$(realpath "$(dirname "$0")/..")
OUT=$REPO/gpurun_out/overlap
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 $REPO/tools/overlap_probe.py ${1:-2} ${2:-8} > $OUT/probe.log 2>&1
cat $OUT/probe.log | tail -3
ls -R $OUT | head -20
