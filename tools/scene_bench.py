"""Scene compositing: GPU path vs the scipy restatement of the reference loop (field_deblender.py:46-97).  GPU only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from oracle import scene_oracle as so

ctx = E.default_context()
rng = np.random.default_rng(0)
F, cs, nb = 259, 59, 6
field = rng.normal(size=(F, F, nb))
for N, frac in ((100, 0.0), (1000, 0.0), (1000, 0.1)):
    stamps = rng.random((N, cs, cs, nb))
    pos = np.rint(rng.uniform(-100, 100, size=(N, 2)))
    k = int(N * frac)
    pos[:k] += rng.uniform(-0.5, 0.5, size=(k, 2))
    ctx.scene_composite(field, stamps[:2], pos[:2], -1.0)
    t0 = time.perf_counter(); got = ctx.scene_composite(field, stamps, pos, -1.0); t1 = time.perf_counter()
    n_cpu = min(N, 20)
    t2 = time.perf_counter(); so.residual_field(field, stamps[:n_cpu], pos[:n_cpu], cs); t3 = time.perf_counter()
    print(f"N={N} sub-pixel {frac:.0%}: GPU (incl. host copies) {1e3 * (t1 - t0):.1f} ms = {N / (t1 - t0):.0f} objects/s; "
          f"scipy loop {1e3 * (t3 - t2) / n_cpu:.1f} ms/object = {n_cpu / (t3 - t2):.1f} objects/s")
cut_pos = rng.integers(0, F - cs, size=(10000, 2))
t0 = time.perf_counter(); c = ctx.scene_extract(field, cut_pos, cs); t1 = time.perf_counter()
print(f"extract 10000 cutouts: {1e3 * (t1 - t0):.1f} ms ({c.nbytes / (t1 - t0) / 1e9:.2f} GB/s of float64 output incl. D2H)")
