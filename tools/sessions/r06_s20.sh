#!/bin/bash
# round 6, GPU session 20: the two bf16 seam changes again, longer runs (1000 queued steps each), six alternations
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s20
mkdir -p $O
cd $R
python tools/bf16_bench.py 256 1000 1 > /dev/null 2>&1   # warm the box
for rep in 1 2 3 4 5 6; do
  echo -n "round-6 form so far (both off)  "; DV_BF_SUMS_ON_MAIN=1 DV_BF_TAIL_ON_WGRAD_STREAM=1 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "sums on the reduction stream    "; DV_BF_TAIL_ON_WGRAD_STREAM=1 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "tail on the main stream         "; DV_BF_SUMS_ON_MAIN=1 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "both (default)                  "; python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/ab.txt
