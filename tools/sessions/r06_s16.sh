#!/bin/bash
# round 6, GPU session 16: the attribution table of round 5, again, on the round-6 bf16 step (development build, timing only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s16
mkdir -p $O
cd $R
for rep in 1 2; do
  echo -n "unmodified                       "; DV_EXP_SKIP_SMALL=0 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "no weight-gradient kernels        "; DV_EXP_SKIP_WGRAD=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "conv K loops cut to one step      "; DV_EXP_BCONV=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "both                              "; DV_EXP_BCONV=1 DV_EXP_SKIP_WGRAD=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "no conv launches, no wgrad        "; DV_EXP_BCONV=2 DV_EXP_SKIP_WGRAD=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  echo -n "conv = empty workgroups, no wgrad "; DV_EXP_BCONV=6 DV_EXP_SKIP_WGRAD=1 python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
done | tee $O/attribution.txt
