"""Summary of a parity-margins file (tests/margins.py): per recorded case the number of tensors, the largest engine error, the
tensors whose bound is above 1e-3 (gradients whose float32 evaluation is itself further than that from float64) and the
smallest margin bound / error.  usage: margins_summary.py profiles/r05_parity_margins.txt"""
import sys

cases, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("#   "):                      # the explanation / column header lines of a case
        continue
    if ln.startswith("# "):
        cur = {"name": ln[2:].strip(), "rows": []}
        cases.append(cur)
    elif ln.startswith("    ") and cur is not None:
        f = ln.split()
        try:
            name, eng = f[0], float(f[1])
            other = None if f[2] == "-" else float(f[2])
            bound = float(f[3])
        except (ValueError, IndexError):
            continue
        cur["rows"].append((name, eng, other, bound, " ".join(f[4:])))
for c in cases:
    rows = c["rows"]
    if not rows:
        continue
    worst = max(rows, key=lambda r: r[1] / r[3] if r[3] > 0 else 0)
    loose = [r for r in rows if "bound above 1e-3" in r[4]]
    print(f"{c['name']}\n    {len(rows)} tensors; closest to its bound: {worst[0]} {worst[1]:.3e} of {worst[3]:.3e}"
          f" ({worst[1] / worst[3]:.2f}); largest error {max(r[1] for r in rows):.3e}")
    for r in loose:
        print(f"    bound above 1e-3: {r[0]:32s} engine {r[1]:.3e}  numpy-float32 {r[2]:.3e}  bound {r[3]:.3e}")
