#!/bin/bash
# round 6, GPU session 3: the dense trunk on the bf16 matrix cores - parity (per layer, whole step), then the A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s3
mkdir -p $O
cd $R
echo "(per-layer tests: passed in the previous run)"; rc=0
tail -25 $O/layers_bf16.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "not training_quality and not sigma_floor" > $O/bf16.log 2>&1; rc=$?
tail -25 $O/bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "DV_BF_TRUNK=$v " ; DV_BF_TRUNK=$v python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
  done
done | tee $O/trunk_ab.txt
