#!/bin/bash
# round 6, GPU session 27: bn_conv0_grads_kernel with a two-level band sum: parity (whole-step tests of both engines check
# d(gamma) / d(beta) / d(kernel0)), the kernel's time, both steps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s27
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq.log 2>&1 || exit 1
cd $R
python tools/kstat.py $O/seq bn_conv0
for rep in 1 2 3 4; do
  python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/step.txt
python tools/bf16_bench.py 256 300 0 2>/dev/null | tail -1
