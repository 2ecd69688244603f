#!/bin/bash
# round 6, GPU session 7: fused sampler-side backward of the trunk: parity, determinism, timing
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s7
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_0_layers_bf16.py -x -q -m gpu -k "59px-256-2 or 59px-48-2 or 128px-16-2 or 29px-k55-24" > $O/layers_bf16.log 2>&1; rc=$?
tail -15 $O/layers_bf16.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_0_arch_variants.py -x -q -m gpu -k "not training_quality and not sigma_floor and not quoted and not 128px_at" > $O/bf16.log 2>&1; rc=$?
tail -15 $O/bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "DV_BF_TRUNK=$v " ; DV_BF_TRUNK=$v python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
  done
done | tee $O/trunk_ab.txt
for rep in 1 2; do python tools/bf16_bench.py 256 300 0 2>/dev/null | tail -1; done | tee $O/f32.txt
