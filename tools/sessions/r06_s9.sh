#!/bin/bash
# round 6, GPU session 9: bf16 engine with 8 .. 15 bands
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s9
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_0_arch_variants.py -x -q -m gpu > $O/arch.log 2>&1; rc=$?
tail -25 $O/arch.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "toy or dc2 or inference_matches or reproducible" > $O/bf16.log 2>&1; rc=$?
tail -8 $O/bf16.log
exit $rc
