#!/bin/bash
# round 6, GPU session 10: gate-matched float64 comparison (small case, then the 128-px case at 64 stamps)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s10
mkdir -p $O
cd $R
DV_PARITY_MARGINS=$O/margins.txt timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_0_fullsize_oracle.py tests/test_gpu_0_arch_variants.py -x -q -m gpu -s -k "full_arch_parity_b4 or 128px_six_level or refuses" > $O/gate.log 2>&1; rc=$?
tail -30 $O/gate.log
exit $rc
