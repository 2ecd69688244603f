#!/bin/bash
# round 6, GPU session 23: two-row weight gradient with incremental DMA addresses: parity of the layers, per-launch times, step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s23
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_0_layers_bf16.py -x -q -m gpu > $O/layers.log 2>&1 || { tail -30 $O/layers.log; exit 1; }
tail -2 $O/layers.log
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq2 -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq2.log 2>&1 || exit 1
cd $R
python tools/kstat.py $O/seq2 bwgrad
for rep in 1 2 3 4; do
  echo -n "two-row form  "; python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/ab.txt
