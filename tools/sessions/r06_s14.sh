#!/bin/bash
# round 6, GPU session 14: dense kernel gradients in one pass (fp32 TN kernel): parity, A/B; per-call times of the drop-in deblend_field
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s14
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_0_layers_f32.py tests/test_gpu_bf16.py -x -q -m gpu -k "not 128px-64 and not quoted and not 128px_at and not training_quality and not sigma_floor and not wide_channel" > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then echo -n "f32, tiled dense wgrad: "; DV_DENSE_WGRAD_TILED=1 python tools/bf16_bench.py 256 300 0 2>/dev/null | tail -1
    else echo -n "f32, one-pass dense wgrad: "; python tools/bf16_bench.py 256 300 0 2>/dev/null | tail -1; fi
  done
done | tee $O/dense_wgrad_ab.txt
for rep in 1 2; do
  echo -n "bf16, tiled W0 wgrad: "; DV_DENSE_WGRAD_TILED=1 python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
  echo -n "bf16, one-pass W0 wgrad: "; python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
done | tee -a $O/dense_wgrad_ab.txt
python tools/probes/drop_in_calls.py 0 0 2>&1 | grep call | tee $O/drop_in_calls.txt
