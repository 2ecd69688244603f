"""Matrix-pipe occupancy per kernel family over one training step from a rocprofv3 counter pass:

  cd /tmp && export TMPDIR=/tmp
  DV_NO_OVERLAP=1 DV_NO_FWD_SPLIT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace \\
      --output-format csv -d <dir> -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_mfma.py <dir> > profiles/r01_pmc_mfma_vNN.json

mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(MI355X_MICROARCH.md, DVFS section); v_mfma_f32_16x16x4_f32 holds the pipe for 32 cycles."""
import csv, glob, json, sys

FAMILIES = (("bconv_uni_kernel / bconv_kernel (bf16)", ("bconv_",)), ("bwgrad_kernel / bwgrad2_kernel (bf16)", ("bwgrad_kernel", "bwgrad2_kernel")),
            ("bgemm_kernel / bgemm_tn_kernel (bf16 dense trunk, round 6)", ("bgemm_",)),
            ("wino_conv4_kernel (Winograd F(2x2,3x3) forward / data gradient, four-wave form)", ("wino_conv4_kernel",)),
            ("wino_conv_kernel (same, eight-wave form: the 32-input-channel launches)", ("wino_conv_kernel",)),
            ("wino_wgrad_kernel (Winograd-domain weight gradient)", ("wino_wgrad_kernel",)),
            ("gconv_strip (incl. first-layer form)", ("gconv_strip",)), ("gconv_s2", ("gconv_s2",)),
            ("gconv2 / gconv", ("gconv2_kernel", "gconv_kernel")), ("wgrad_strip (incl. first-layer form)", ("wgrad_strip",)),
            ("wgrad (tiled)", ("wgrad_kernel",)))


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return None


f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
marks = [i for i in ids if "fold_bn_w1" in disp[i]["name"]]
lo, hi = marks[-2], marks[-1]
fams, tot_b, tot_c = {}, 0.0, 0.0
for i in ids:
    if not (lo <= i < hi):
        continue
    d = disp[i]
    fam = family(d["name"])
    if fam is None:
        continue
    e = fams.setdefault(fam, {"launches": 0, "mfma_busy_cycles": 0.0, "kernel_cycles": 0.0})
    e["launches"] += 1
    e["mfma_busy_cycles"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    e["kernel_cycles"] += d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    for extra in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):      # optional wave-level counters
        if extra in d:
            e[extra] = e.get(extra, 0.0) + d[extra]
for e in fams.values():
    e["mfma_pipe_busy"] = round(e["mfma_busy_cycles"] / (1024.0 * e["kernel_cycles"]), 4)
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_any_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 4)
        e["active_inst_of_wave_cycles"] = round(e.get("SQ_ACTIVE_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 4)
    tot_b += e["mfma_busy_cycles"]
    tot_c += e["kernel_cycles"]
print(json.dumps({
    "source": "DV_NO_OVERLAP=1 DV_NO_FWD_SPLIT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace on "
              "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline`, MI355X; one training step; tools/pmc_mfma.py",
    "definition": "mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs "
                  "(MI355X_MICROARCH.md: DVFS section); v_mfma_f32_16x16x4_f32 holds the pipe for 32 cycles",
    "families": fams, "all_matrix_kernels": {"mfma_pipe_busy": round(tot_b / (1024.0 * tot_c), 4)}}, indent=1))
