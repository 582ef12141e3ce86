#!/bin/bash
# GPU box: parity suite, then timing of this tree's library against variants, then lane diagnostics.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; echo "OMP $OMP_NUM_THREADS"; } > gpurun_out/box_facts.txt 2>&1
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 600 bash tools/ab_libs.sh ab_a.txt "$@" || exit 1
timeout -k 10 200 python3 tools/kernel_stats.py 4 > gpurun_out/kstats_a.txt 2>&1 || { tail gpurun_out/kstats_a.txt; exit 1; }
tail -12 gpurun_out/kstats_a.txt
