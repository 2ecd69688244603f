#!/bin/bash
# round 6, GPU session 19: bf16 step seams - loss sums / head bias sums on the reduction stream, the tail of the pass on the
# main stream; parity of the bf16 engine first, then alternating A/B runs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s19
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_0_layers_bf16.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for rep in 1 2 3 4; do
  echo -n "round-6 form so far (both off)  "; DV_BF_SUMS_ON_MAIN=1 DV_BF_TAIL_ON_WGRAD_STREAM=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "sums on the reduction stream    "; DV_BF_TAIL_ON_WGRAD_STREAM=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "tail on the main stream         "; DV_BF_SUMS_ON_MAIN=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "both (default)                  "; python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
done | tee $O/ab.txt
