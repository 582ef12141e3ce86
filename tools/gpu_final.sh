#!/bin/bash
# GPU box: the whole GPU suite, the headline bench line, short bench lines of the other configurations, the 2-rank rehearsal.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out/final
timeout -k 10 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/final/gputests.log 2>&1 || { tail -30 gpurun_out/final/gputests.log; exit 1; }
tail -3 gpurun_out/final/gputests.log
timeout -k 10 400 python3 bench.py > gpurun_out/final/bench_cfg4.json 2> gpurun_out/final/bench_cfg4.err || { tail gpurun_out/final/bench_cfg4.err; exit 1; }
for c in 2 3 5; do S=30; [ $c = 5 ] && S=3; timeout -k 10 400 python3 bench.py --config $c --steps $S --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_cfg$c.json 2> gpurun_out/final/bench_cfg$c.err || { tail gpurun_out/final/bench_cfg$c.err; exit 1; }; done
PT_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/final/bench_2rank_rehearsal.json 2> gpurun_out/final/bench_2rank.err || { tail gpurun_out/final/bench_2rank.err; exit 1; }
python3 - <<'PY'
import json
for n in ("cfg4","cfg2","cfg3","cfg5","2rank_rehearsal"):
    d=json.load(open("gpurun_out/final/bench_%s.json"%n))
    r=d.get("roofline",{})
    print(n, "%.1f Msamples/s"%d["value"], "%.2f ms/step"%d["ms_per_step"], "frac", r.get("frac"), "peak", r.get("peak"), "achieved", r.get("achieved"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
