#!/usr/bin/env python3
"""Generates the numbers block of profiles/README.md from the committed summaries themselves (bench line JSON, rocprofv3
kernel statistics CSV, PMC traffic / MFMA-busy JSON), so that the index cannot disagree with the files it indexes
(the hand-written row said wino_wgrad busy 0.47 where the committed JSON said 0.277).

    python tools/profiles_readme.py            # rewrites the block between the markers in profiles/README.md
    python tools/profiles_readme.py --check    # exit 1 if the committed block differs (tests/test_profiles_readme.py)
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
BEGIN, END = "<!-- BEGIN GENERATED (tools/profiles_readme.py) -->", "<!-- END GENERATED -->"
FP32_PEAK = 157.3


def _load(name):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def _kernel_stats(name, top=8):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        return None
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            n = r["Name"].replace("void ", "").replace("dv::", "")
            n = n.split("(")[0]
            rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
    rows.sort(key=lambda t: -t[3])
    return rows[:top]


def round_block(tag):
    out = []
    line = _load(f"{tag}_bench_line.json")
    if line:
        rf = line.get("roofline") or {}
        out.append(f"### {tag}: `{tag}_bench_line.json`")
        out.append("")
        out.append(f"* headline: **{line['value']:.0f} {line['unit']}**, {line['ms_per_step']:.3f} ms/step, dtype {line['dtype']}, "
                   f"{line['steps']} steps, `{line['config']['workload'][:60]}...`")
        if rf:
            ex = rf.get("executed") or {}
            parts = [f"roofline kernel `{rf.get('kernel')}`: frac {rf.get('frac'):.3f} of {rf.get('peak')} {rf.get('unit')}"]
            if rf.get("algorithmic_frac") is not None:
                parts.append(f"algorithmic frac {rf['algorithmic_frac']:.3f}")
            elif ex:
                parts.append(f"(algorithmic, as that round priced it; executed {ex.get('frac'):.3f})")
            if rf.get("whole_step_frac") is not None:
                parts.append(f"whole step {rf['whole_step_tflops']:.1f} TFLOP/s = {rf['whole_step_frac']:.3f}")
            if rf.get("traffic"):
                parts.append(f"traffic {rf['traffic'] / 1e9:.2f} GB")
            out.append("* " + "; ".join(parts))
            out.append("")
            out.append("| kernel family (live HIP events, streams serialised) | launches/step | ms/step | avg µs | TFLOP/s algorithmic | executed frac |")
            out.append("|---|---|---|---|---|---|")
            for k in rf.get("kernels", []):
                exf = k.get("executed_frac", k.get("frac"))
                out.append(f"| `{k['kernel']}` | {k['launches_per_step']:.0f} | {k['ms_per_step']:.3f} | {k['avg_us']:.1f} | "
                           f"{k['tflops']:.1f} | {exf:.3f} |")
        cb = line.get("cpu_baseline")
        if cb and cb.get("value"):
            out.append("")
            out.append(f"* cpu_baseline ({cb['kind']}): {cb['value']:.0f} {cb['unit']} on {cb['cores']} cores")
        sec = line.get("secondary") or {}
        for name, e in sec.items():
            if isinstance(e, dict) and e.get("value"):
                extra = f", {e['ms_per_step']:.3f} ms/step" if e.get("ms_per_step") else ""
                out.append(f"* secondary `{name}`: {e['value']:.0f} {e.get('unit', '')}{extra}")
        out.append("")
    for suffix, title in (("bench_kernel_stats_sequential.csv", "fp32 step, streams serialised"),
                          ("bf16_kernel_stats_sequential.csv", "bf16 step, streams serialised")):
        ks = _kernel_stats(f"{tag}_{suffix}")
        if ks:
            out.append(f"### {tag}: `{tag}_{suffix}` ({title}; rocprofv3 --kernel-trace --stats, top kernels by total time)")
            out.append("")
            out.append("| kernel | calls | avg µs | total ms | % |")
            out.append("|---|---|---|---|---|")
            for n, c, a, t, p in ks:
                out.append(f"| `{n}` | {c} | {a:.1f} | {t:.3f} | {p:.2f} |")
            out.append("")
    for fname, title in ((f"{tag}_pmc_f32_mfma.json", "fp32 step"), (f"{tag}_pmc_bf16_mfma.json", "bf16 step")):
        d = _load(fname)
        if d:
            out.append(f"### {tag}: `{fname}` (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), {title})")
            out.append("")
            out.append("| kernel family | launches | matrix pipe busy | waves waiting | waves issuing |")
            out.append("|---|---|---|---|---|")
            for fam, v in d["families"].items():
                out.append(f"| {fam} | {v['launches']} | {v['mfma_pipe_busy']:.3f} | {v.get('wait_any_of_wave_cycles', float('nan')):.3f} | "
                           f"{v.get('active_inst_of_wave_cycles', float('nan')):.3f} |")
            out.append(f"| **all matrix kernels** | | **{d['all_matrix_kernels']['mfma_pipe_busy']:.3f}** | | |")
            out.append("")
    d = _load(f"{tag}_pmc_traffic.json")
    if d:
        out.append(f"### {tag}: `{tag}_pmc_traffic.json` (HBM bytes of one step: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes)")
        out.append("")
        for key, title in (("per_step_bytes", "fp32 step"), ("bf16_per_step_bytes", "bf16 step")):
            if key in d:
                out.append(f"| kernel family ({title}) | GB per step |")
                out.append("|---|---|")
                for fam, v in d[key].items():
                    if fam != "total":
                        out.append(f"| {fam} | {v['hbm_bytes'] / 1e9:.3f} |")
                out.append(f"| **total** | **{d[key]['total'] / 1e9:.3f}** |")
                out.append("")
    return out


def generate():
    tags = sorted({f.split("_")[0] for f in os.listdir(PROF) if f[:1] == "r" and f[1:3].isdigit() and "_" in f}, reverse=True)
    lines = [BEGIN, "",
             "## Numbers of the committed summaries (generated from the files themselves; do not edit by hand)", ""]
    for tag in tags:
        if int(tag[1:]) >= 3:
            lines += round_block(tag)
    lines.append(END)
    return "\n".join(lines)


def main():
    path = os.path.join(PROF, "README.md")
    txt = open(path).read()
    block = generate()
    if BEGIN in txt:
        new = txt[:txt.index(BEGIN)] + block + txt[txt.index(END) + len(END):]
    else:
        head, _, rest = txt.partition("\n")
        new = head + "\n\n" + block + "\n" + rest
    if "--check" in sys.argv[1:]:
        if new != txt:
            print("profiles/README.md: the generated block is stale - run python tools/profiles_readme.py", file=sys.stderr)
            return 1
        return 0
    with open(path, "w") as fh:
        fh.write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main())
