"""Per-launch durations of the bf16 conv kernels over one serialised training step, from a rocprofv3 kernel trace:
  DV_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/bf16_bench.py 256 3
  python tools/bconv_launches.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fold_bn_w1" in r["Kernel_Name"]]
seg = rows[marks[-2]:marks[-1]]
tot = 0.0
out = []
for r in seg:
    n = r["Kernel_Name"]
    if "bconv" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        short = n.split("dv::")[1].split("(")[0]
        out.append(f"{short}:{r['Grid_Size_X']}:{d:.1f}")
print(" ".join(out))
print("bconv total us", round(tot, 1), "step us", (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3)
