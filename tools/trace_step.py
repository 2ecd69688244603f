"""Per-kernel timeline of one train step from a rocprofv3 kernel trace csv: trace_step.py <kernel_trace.csv> [filter]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("dv::", "").replace("(anonymous namespace)::", "")[:70]
    if flt in n:
        print(f"{d:8.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} x{r['Workgroup_Size_X']:>4s}  {n}")
print("sum of kernel times", round(tot, 1), "us; wall", (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
