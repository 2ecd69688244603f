#!/bin/bash
# round 6, GPU session 24: timeline of the bf16 step at the final build (overlapped streams)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s24
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/ovl -o t -- python3 $R/tools/bf16_bench.py 256 6 > $O/ovl.log 2>&1 || exit 1
ls -la $O/ovl
