#!/bin/bash
# round 6, GPU session 18: upper bounds for the serial seams of a train step (development build, timing only):
# DV_EXP_SKIP_TAIL bits - 1 shallow-bucket tail, 2 loss sums + head bias sums on the main stream, 4 (fp32) bn_finalize + bn_apply
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s18
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    echo -n "bf16 SKIP_TAIL=$v " ; DV_EXP_SKIP_TAIL=$v timeout -k 10 120 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  done
  for v in 0 1 2 4 7; do
    echo -n "fp32 SKIP_TAIL=$v " ; DV_EXP_SKIP_TAIL=$v timeout -k 10 120 python tools/bf16_bench.py 256 200 0 2>/dev/null | tail -1
  done
done | tee $O/skip_tail.txt
