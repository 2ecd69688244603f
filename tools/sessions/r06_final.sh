#!/bin/bash
# round 6: what the driver runs at round end - the GPU suite, smoke(), the default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
DV_PARITY_MARGINS=$O/parity_margins.txt python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1; rc=$?
tail -25 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final/bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, d["roofline"]["frac"])
for k, v in d["secondary"].items():
    print(k, v.get("value"), v.get("ms_per_step"))
PY
