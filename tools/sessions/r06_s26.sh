#!/bin/bash
# round 6, GPU session 26: bf_cast_kernel - every transposed kernel through the LDS-tiled path: parity (the whole bf16 suite
# reads these matrices), the kernel's launch times, the step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s26
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_0_arch_variants.py tests/test_gpu_0_layers_bf16.py -x -q -m gpu > $O/bf16.log 2>&1 || { tail -30 $O/bf16.log; exit 1; }
tail -2 $O/bf16.log
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 5 > $O/seq.log 2>&1 || exit 1
cd $R
python tools/kstat.py $O/seq bf_cast
for rep in 1 2 3 4; do
  python tools/bf16_bench.py 256 1000 1 2>/dev/null | tail -1
done | tee $O/step.txt
