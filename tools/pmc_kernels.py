"""Per-kernel SQ counters from a rocprofv3 counter pass (any program): busy fraction of the matrix pipe, where the waves
wait, LDS bank conflicts.
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \\
      --kernel-trace --output-format csv -d <dir> -o m -- python3 tools/wino_check.py benchonly
  python tools/pmc_kernels.py <dir> [name filter]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    key = (n.split("(")[0][-60:], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    a = agg.setdefault(key, {})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["_n"] = a.get("_n", 0) + (1 if r["Counter_Name"] == "GRBM_GUI_ACTIVE" else 0)
for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc <= 0:
        continue
    wc = a.get("SQ_WAVE_CYCLES", 0)
    out = f"{n:60s} grid {g:>9s} x{a['_n']:4d}  cycles/launch {cyc / max(1, a['_n']):9.0f}  mfma busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc):5.3f}"
    if wc:
        out += f"  wait_any {a.get('SQ_WAIT_ANY', 0) / wc:5.3f}  wait_inst {a.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f}  active {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f}"
    if "SQ_LDS_IDX_ACTIVE" in a:
        out += f"  lds_active {a['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):5.3f} conflict {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, a['SQ_LDS_IDX_ACTIVE']):5.3f}"
    print(out)
