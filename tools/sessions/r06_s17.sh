#!/bin/bash
# round 6, GPU session 17: what a cross-step pipeline of the decoder's late kernel gradients could gain (fp32 engine;
# development build, timing only: DV_EXP_DEFER_WGRAD=n holds the head conv's and the last n transposed convs' kernel
# gradients back and queues them on the weight-gradient stream at the start of the next forward pass)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s17
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in 0 1 2 3 4 6; do
    echo -n "DEFER_WGRAD=$v " ; DV_EXP_DEFER_WGRAD=$v timeout -k 10 120 python tools/bf16_bench.py 256 200 0 2>/dev/null | tail -1
  done
done | tee $O/f32_defer_wgrad.txt
