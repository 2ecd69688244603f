#!/bin/bash
# round 6, GPU session 4: kernel statistics of the bf16 step with the trunk on the matrix cores
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq -o s -- python3 $R/tools/bf16_bench.py 256 10 1 > $O/seq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/ovl -o t -- python3 $R/tools/bf16_bench.py 256 6 1 > $O/ovl.log 2>&1 || exit 1
find $O -name "*.csv" | xargs ls -la
