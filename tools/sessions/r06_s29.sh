#!/bin/bash
# round 6, GPU session 29: the head-carrying row kernel with 4-pixel strips (DV_BF_HEAD_STRIP=4) against 8: direct test, times
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s29
mkdir -p $O
cd $R
DV_BF_HEAD_STRIP=4 timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "head_in_the_head" > $O/head.log 2>&1 || { tail -30 $O/head.log; exit 1; }
tail -2 $O/head.log
cd /tmp && export TMPDIR=/tmp
DV_BF_HEAD_STRIP=4 DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq.log 2>&1 || exit 1
cd $R
python tools/kstat.py $O/seq bconv_row
for rep in 1 2 3; do
  echo -n "8-pixel strips  "; python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
  echo -n "4-pixel strips  "; DV_BF_HEAD_STRIP=4 python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/ab.txt
