#!/bin/bash
# the whole GPU suite as the driver runs it, with durations and the parity margins
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_full
mkdir -p $O
cd $R
DV_PARITY_MARGINS=$O/parity_margins.txt python -m pytest tests -x -q -m gpu --durations=25 > $O/gpu_tests.log 2>&1
rc=$?
tail -45 $O/gpu_tests.log
exit $rc
