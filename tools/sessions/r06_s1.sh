#!/bin/bash
# round 6, GPU session 1: baseline of the new build, pricing of the fp32 weight-gradient families, traces for the timelines
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s1
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
tail -c 600 $O/bench_default.json
echo "== fp32: weight-gradient families left out (development build; 0 = all in, 1 = tiled, 3 = Winograd-domain, 4 = strip, 2 = all)"
for rep in 1 2 3; do
  for v in 0 1 3 4 2; do
    echo -n "SKIP_WGRAD=$v " ; DV_EXP_SKIP_WGRAD=$v python tools/bf16_bench.py 256 200 0 2>/dev/null | tail -1
  done
done | tee $O/f32_skip_wgrad.txt
echo "== fp32: wino_wgrad workgroups per launch"
for rep in 1 2 3; do
  for v in 128 192 256; do
    echo -n "WINOW_WGS=$v " ; DV_EXP_WINOW_WGS=$v python tools/bf16_bench.py 256 200 0 2>/dev/null | tail -1
  done
done | tee $O/f32_winow_wgs.txt
echo "== bf16: trunk launches left out (0 / 1 = nine small neighbours / 2 = the whole dense trunk)"
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo -n "SKIP_SMALL=$v " ; DV_EXP_SKIP_SMALL=$v python tools/bf16_bench.py 256 200 1 2>/dev/null | tail -1
  done
done | tee $O/bf16_skip_small.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/f32_ovl -o t -- python3 $R/tools/bf16_bench.py 256 6 0 > $O/f32_ovl.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/bf16_ovl -o t -- python3 $R/tools/bf16_bench.py 256 6 1 > $O/bf16_ovl.log 2>&1 || exit 1
find $O -name "*.csv" | xargs ls -la
