#!/bin/bash
# GPU box: tools/profile.sh for the configurations given (default 4 2 3 5), results under gpurun_out/prof_cfg<N>/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for C in ${@:-4 2 3 5}; do
  timeout -k 10 900 bash tools/profile.sh prof_cfg$C --config $C > gpurun_out/prof_cfg$C.log 2>&1 || { tail -20 gpurun_out/prof_cfg$C.log; exit 1; }
  grep -E "rc=|path_kernel  " gpurun_out/prof_cfg$C.log | head -12
done
