#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the DEFAULT bench command (its steps are enqueued back to back, so launches overlap).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trace_default; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $R/bench.py --no-cpu-baseline --no-count --no-exact > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cat $OUT/p_kernel_stats.csv
python3 - $OUT/p_kernel_trace.csv <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'path_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
print("path_kernel launches: queue, start ms, end ms, duration ms")
for r in rows:
    print(r['Queue_Id'], "%.2f %.2f %.2f" % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
ends=[int(r['End_Timestamp']) for r in rows[3:]]
print("mean time between the ends of consecutive launches (timed region): %.2f ms" % ((ends[-1]-ends[0])/1e6/(len(ends)-1)))
PY
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print('bench line: %.1f Msamples/s, %.2f ms/step, avg_launch_ms %.2f, wall_ms_per_launch %.2f, frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['wall_ms_per_launch'], r['frac']))"
