#!/bin/bash
# Collects the rocprofv3 evidence of a round on an MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03
# kernel-trace statistics of the fp32 headline and of the bf16 train step (streams serialised and overlapped), and the
# PMC passes (HBM traffic: FETCH_SIZE and WRITE_SIZE in separate runs; matrix-pipe occupancy) - counters always in runs
# of their own with --kernel-trace only, the program itself (python3 ...) right behind "--".
set -o pipefail
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary"
F="python3 $R/tools/bf16_bench.py 256 5"
DV_NO_OVERLAP=1 DV_NO_FWD_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32_seq -o s -- $B > $O/f32_seq.log 2>&1 || exit 1
echo "fp32 sequential done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32_ovl -o s -- $B > $O/f32_ovl.log 2>&1 || exit 1
echo "fp32 overlapped done"
DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_seq -o s -- $F > $O/bf16_seq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_ovl -o s -- $F > $O/bf16_ovl.log 2>&1 || exit 1
echo "bf16 traces done"
P="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f32/fetch -o f -- $P > $O/pmc_f32_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_f32/write -o w -- $P > $O/pmc_f32_write.log 2>&1 || exit 1
echo "fp32 traffic done"
G="python3 $R/tools/bf16_bench.py 256 3"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_bf16/fetch -o f -- $G > $O/pmc_bf16_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_bf16/write -o w -- $G > $O/pmc_bf16_write.log 2>&1 || exit 1
echo "bf16 traffic done"
DV_NO_OVERLAP=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_bf16_mfma -o m -- $G > $O/pmc_bf16_mfma.log 2>&1 || exit 1
echo "bf16 MFMA counters done"
DV_NO_OVERLAP=1 DV_NO_FWD_SPLIT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_f32_mfma -o m -- $P > $O/pmc_f32_mfma.log 2>&1 || exit 1
echo "fp32 MFMA counters done"
cd $R
python3 tools/pmc_traffic.py $O/pmc_f32 per_step_bytes > $O/pmc_traffic_f32.json
python3 tools/pmc_traffic.py $O/pmc_bf16 bf16_per_step_bytes > $O/pmc_traffic_bf16.json
python3 tools/pmc_mfma.py $O/pmc_f32_mfma > $O/pmc_f32_mfma.json
python3 tools/pmc_mfma.py $O/pmc_bf16_mfma > $O/pmc_bf16_mfma.json
python3 tools/pmc_kernels.py $O/pmc_f32_mfma > $O/pmc_f32_kernels.txt
# the files that go to profiles/ (small: statistics csv + JSON summaries; the raw traces stay in gpurun_out)
S=$O/summary
mkdir -p $S
python3 - "$O" "$S" "$TAG" <<'PY'
import glob, json, shutil, sys
O, S, TAG = sys.argv[1:4]
f32, bf = json.load(open(O + "/pmc_traffic_f32.json")), json.load(open(O + "/pmc_traffic_bf16.json"))
f32["bf16_source"] = bf["source"]
f32["bf16_per_step_bytes"], f32["bf16_per_kernel_bytes"] = bf["bf16_per_step_bytes"], bf["bf16_per_kernel_bytes"]
json.dump(f32, open(f"{S}/{TAG}_pmc_traffic.json", "w"), indent=1)
shutil.copy(O + "/pmc_f32_mfma.json", f"{S}/{TAG}_pmc_f32_mfma.json")
shutil.copy(O + "/pmc_bf16_mfma.json", f"{S}/{TAG}_pmc_bf16_mfma.json")
shutil.copy(O + "/pmc_f32_kernels.txt", f"{S}/{TAG}_pmc_f32_kernels.txt")
for src, dst in (("f32_seq", "bench_kernel_stats_sequential"), ("f32_ovl", "bench_kernel_stats_overlapped"),
                 ("bf16_seq", "bf16_kernel_stats_sequential"), ("bf16_ovl", "bf16_kernel_stats_overlapped")):
    hits = glob.glob(f"{O}/{src}/**/*kernel_stats.csv", recursive=True)
    if hits:
        shutil.copy(hits[0], f"{S}/{TAG}_{dst}.csv")
PY
# keep what is judged small: statistics csv + the JSON summaries (the raw traces stay in gpurun_out)
ls -la $O
