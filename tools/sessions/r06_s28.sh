#!/bin/bash
# round 6, GPU session 28: the head in the head conv's epilogue (BEPI_HEAD): the direct test, the bf16 suite, launch times, A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s28
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_in_the_head" > $O/head.log 2>&1 || { tail -30 $O/head.log; exit 1; }
tail -2 $O/head.log
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_0_arch_variants.py -x -q -m gpu > $O/bf16.log 2>&1 || { tail -30 $O/bf16.log; exit 1; }
tail -2 $O/bf16.log
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq.log 2>&1 || exit 1
cd $R
python tools/kstat.py $O/seq bf_head bconv_row
for rep in 1 2 3 4; do
  echo -n "head kernel  "; DV_BF_HEAD_FUSED=0 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "head fused   "; python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/ab.txt
