#!/bin/bash
# round 6, GPU session 21: the cross-step experiment of session 17 on the bf16 engine (development build, timing only:
# DV_EXP_DEFER_WGRAD=n holds the head conv's and the last n transposed convs' kernel gradients back and queues them on the
# weight-gradient stream at the start of the next forward pass)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s21
mkdir -p $O
cd $R
DV_EXP_DEFER_WGRAD=0 python tools/bf16_bench.py 256 1000 1 > /dev/null 2>&1
for rep in 1 2 3 4; do
  for v in 0 1 2 3 4 8; do
    echo -n "bf16 DEFER_WGRAD=$v " ; DV_EXP_DEFER_WGRAD=$v timeout -k 10 120 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  done
done | tee $O/bf16_defer_wgrad.txt
