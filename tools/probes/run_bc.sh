set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
export DV_NO_OVERLAP=1
for pr in 0 1; do
export DV_BCONV_PROBE=$pr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bc_prof$pr -o bc -- python3 tools/bf16_bench.py 256 10 1 > gpurun_out/bc_prof$pr.log 2>&1
done
python3 - <<'PY'
import csv, glob
cols = []
for pr in (0, 1):
    f = glob.glob(f"gpurun_out/bc_prof{pr}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "bconv_" in r["Kernel_Name"]]
    cols.append([(r["Kernel_Name"][9:45], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-33:]])
for i in range(33):
    print("%2d %-38s full %7.1f  no-loop %7.1f" % (i, cols[0][i][0], cols[0][i][1], cols[1][i][1]))
print("totals", [sum(x[1] for x in c) for c in cols])
PY
