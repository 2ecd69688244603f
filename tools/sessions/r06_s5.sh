#!/bin/bash
# round 6, GPU session 5: faster trunk kernels - parity of the trunk products, then timing and kernel statistics
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s5
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_0_layers_bf16.py -x -q -m gpu -k "59px-256-2 or 59px-48-2 or 128px-16-2 or 29px-k55-24" > $O/layers_bf16.log 2>&1; rc=$?
tail -15 $O/layers_bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "DV_BF_TRUNK=$v " ; DV_BF_TRUNK=$v python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
  done
done | tee $O/trunk_ab.txt
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 10 1 > $O/seq.log 2>&1 || exit 1
