"""Fused head (BEPI_HEAD) against the head kernel on one engine: python tools/probes/head_fuse_probe.py B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x, y = synthetic_stamps(B, seed=41)
eng = E.Engine(E.make_config(max_batch=B, dtype=1))
eng.init(seed=6)
eng.upload(0, x, y)
eng.keep_outputs(True)
ref = eng.grad_step(0, first=0, B=B, seed=13)
g0 = {n: eng.get_grad(n).copy() for n, _, tr in eng.specs if tr}
eng.keep_outputs(False)
out = eng.grad_step(0, first=0, B=B, seed=13)
print("kernel:", ref)
print("fused: ", out)
bad = [n for n, g in g0.items() if not np.array_equal(eng.get_grad(n), g)]
print("gradients that differ:", len(bad), bad[:6])
for n in bad[:3]:
    a, b = eng.get_grad(n).astype(np.float64), g0[n].astype(np.float64)
    print(n, np.abs(a - b).max(), np.abs(b).max())
