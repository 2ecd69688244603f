#!/bin/bash
# round 6, GPU session 13: next step's input normalised ahead on the comm stream (bf16): parity / determinism / fit, then A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s13
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_api.py -x -q -m gpu -k "reproducible or bucket or toy_arch_64 or fit or train or queued or config0 or collective or training_quality" > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then echo -n "input at the head of the step: "; DV_NO_INPUT_AHEAD=1 python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
    else echo -n "input ahead on the comm stream: "; python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1; fi
  done
done | tee $O/input_ab.txt
