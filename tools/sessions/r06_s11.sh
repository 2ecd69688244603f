#!/bin/bash
# round 6, GPU session 11: bucket-boundary sums on the reduction stream: determinism / parity, then the A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s11
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "reproducible or bucket or toy_arch_64 or full_arch_dc2 or thousand or full_arch_64 or deep_arch_128px_six" > $O/bf16.log 2>&1; rc=$?
tail -12 $O/bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then echo -n "boundary sums on the weight-gradient stream: "; DV_BF_BOUNDARY_ON_WGRAD_STREAM=1 python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
    else echo -n "boundary sums on the reduction stream:       "; python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1; fi
  done
done | tee $O/boundary_ab.txt
