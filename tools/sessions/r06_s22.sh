#!/bin/bash
# round 6, GPU session 22: bf16 weight gradient - accumulators pinned to the accumulation registers (both forms) and the
# two-row form for stride-1 layers: per-layer parity first, then whole steps, then per-launch times and the step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s22
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_0_layers_bf16.py -x -q -m gpu > $O/layers.log 2>&1 || { tail -30 $O/layers.log; exit 1; }
tail -2 $O/layers.log
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_0_arch_variants.py -x -q -m gpu > $O/bf16.log 2>&1 || { tail -30 $O/bf16.log; exit 1; }
tail -2 $O/bf16.log
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq2 -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq2.log 2>&1 || exit 1
DV_BWGRAD_ONE_ROW=1 DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq1 -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq1.log 2>&1 || exit 1
cd $R
echo "two-row form:"; python tools/kstat.py $O/seq2 bwgrad
echo "one-row form (accumulators pinned):"; python tools/kstat.py $O/seq1 bwgrad
for rep in 1 2 3 4; do
  echo -n "one-row form  "; DV_BWGRAD_ONE_ROW=1 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "two-row form  "; python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/ab.txt
