#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/ovl -o t -- python3 $R/tools/bf16_bench.py 256 8 1 > $O/ovl.log 2>&1 || exit 1
