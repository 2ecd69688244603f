#!/bin/bash
# round 6, GPU session 2: the whole GPU suite with durations (where do its 560 s go), result-pool A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s2
mkdir -p $O
cd $R
python tools/probes/pool_ab.py 0 2>&1 | tee $O/pool_ab_f32.txt || exit 1
python -m pytest tests -x -q -m gpu --durations=60 > $O/gpu_tests.log 2>&1
rc=$?
tail -80 $O/gpu_tests.log
exit $rc
