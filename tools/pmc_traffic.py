"""HBM bytes per training step and kernel family from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; collected
separately, as the guide prescribes):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_traffic.py <dir> > profiles/r01_pmc_traffic_vNN.json

One step = the dispatches between two fold_bn_w1_kernel launches (one per step in both engines).  FETCH_SIZE is doubled (gfx950 correction for wide
coalesced reads, MI355X_MICROARCH.md HBM section); both counters are in KB."""
import csv, glob, json, sys

FAMILIES = (("bconv_kernel / bconv_uni_kernel", ("bconv_",)),
            ("bwgrad_kernel / bwgrad2_kernel", ("bwgrad_kernel", "bwgrad2_kernel")),
            ("bgemm_kernel / bgemm_tn_kernel (dense trunk)", ("bgemm_",)),
            ("wino_wgrad family (wino_wgrad, presum, finish)", ("wino_wgrad",)),
            ("wino_conv_kernel (+ wino_weights_kernel)", ("wino_conv", "wino_weights")),
            ("gconv family (gconv2, gconv_s2, gconv_strip, gconv_strip8, gconv, splitk_finish)", ("gconv", "splitk_finish")),
            ("wgrad family (wgrad, wgrad_strip, wgrad_strip8, reduce_partials)", ("wgrad", "reduce_partials")),
            ("prelu_bwd_kernel", ("prelu_bwd",)))


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def one_pass(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "fold_bn_w1" in r["Kernel_Name"]]
    lo, hi = marks[-2], marks[-1]                  # the last complete step
    out, per = {}, {}
    for r in rows[lo:hi]:
        out[family(r["Kernel_Name"])] = out.get(family(r["Kernel_Name"]), 0.0) + float(r["Counter_Value"])
        base = short_name(r["Kernel_Name"])
        e = per.setdefault(base, [0, 0.0])
        e[0] += 1
        e[1] += float(r["Counter_Value"])
    return out, hi - lo, per


def short_name(name):
    """rocprof kernel name without the namespace, template arguments and signature: `gconv2_kernel`"""
    n = name.split("(")[0].split("<")[0].strip()
    n = n.split(" ")[-1]
    return n.split("::")[-1]


root = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else "per_step_bytes"     # "bf16_per_step_bytes" for the bf16 engine's passes
fetch, nf, kfetch = one_pass(root + "/fetch", "FETCH_SIZE")
write, nw, kwrite = one_pass(root + "/write", "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on `python bench.py --steps 3 "
                 "--warmup 1 --no-cpu-baseline --no-roofline`, MI355X; one training step (between two fold_bn_w1_kernel "
                 f"dispatches: {nf} / {nw} dispatches); tools/pmc_traffic.py",
       "correction": "FETCH_SIZE doubled as prescribed for gfx950 wide coalesced reads (MI355X_MICROARCH.md, HBM section); units KB",
       label: {}}
tot = 0.0
for fam in sorted(set(fetch) | set(write)):
    fr, wr = fetch.get(fam, 0.0), write.get(fam, 0.0)
    res[label][fam] = {"fetch_raw_kb": fr, "write_kb": wr, "hbm_bytes": (2.0 * fr + wr) * 1024.0}
    tot += (2.0 * fr + wr) * 1024.0
res[label]["total"] = tot
# the same per rocprof kernel name (launches of one step)
kl = label.replace("per_step_bytes", "per_kernel_bytes")
res[kl] = {}
for k in sorted(set(kfetch) | set(kwrite)):
    n, fr = kfetch.get(k, [0, 0.0])
    n2, wr = kwrite.get(k, [0, 0.0])
    res[kl][k] = {"launches": max(n, n2), "fetch_raw_kb": fr, "write_kb": wr, "hbm_bytes": (2.0 * fr + wr) * 1024.0}
print(json.dumps(res, indent=1))
