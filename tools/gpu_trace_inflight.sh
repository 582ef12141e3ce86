#!/bin/bash
# GPU box: rocprofv3 kernel trace of one rank's share of a sharded render (pixel tiles, 1 of 8), first with a wait after every
# render, then with the renders in flight and a second output per render: who runs when.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trace_inflight; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in wait flight+second-output; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $M -- python3 $R/tools/flight_probe.py --modes=$M 4 8 > $OUT/$M.txt 2> $OUT/$M.err || { tail $OUT/$M.err; exit 1; }
  grep split $OUT/$M.txt
  python3 - $OUT/${M}_kernel_trace.csv <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'path_kernel' in r['Kernel_Name'] or 'resolve' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-16:]
t0=int(rows[0]['Start_Timestamp'])
print("last launches: kernel, queue, start ms, end ms, duration ms")
for r in rows:
    print("%-8s %s %.2f %.2f %.2f" % (r['Kernel_Name'][:12].replace('void ',''), r['Queue_Id'], (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
done
