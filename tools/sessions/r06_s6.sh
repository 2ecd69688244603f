#!/bin/bash
# round 6, GPU session 6: per-bucket casts on the comm stream + sampler slab sum: determinism / bucket tests, then timing
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s6
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "reproducible or bucket or toy_arch_64 or full_arch_dc2 or inference_matches or thousand" > $O/bf16.log 2>&1; rc=$?
tail -15 $O/bf16.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "DV_BF_TRUNK=$v " ; DV_BF_TRUNK=$v python tools/bf16_bench.py 256 300 1 2>/dev/null | tail -1
  done
done | tee $O/trunk_ab.txt
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 10 1 > $O/seq.log 2>&1 || exit 1
