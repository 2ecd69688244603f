"""One-stamp forward timing of the cooperative layer-stack kernels (run with DV_SMALL_WGS_PER_CU / DV_SMALL_DBG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
eng = E.Engine(E.make_config(max_batch=32))
eng.init(seed=1)
x, _ = synthetic_stamps(8, seed=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for small in (8, 0):
    eng.set_small_forward(small)
    for n in (1, 8):
        for _ in range(5):
            eng.encode(x[:n])
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.encode(x[:n])
        te = (time.perf_counter() - t0) / reps
        for _ in range(5):
            eng.infer(x[:n])
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.infer(x[:n])
        ti = (time.perf_counter() - t0) / reps
        print(f"small_max={small} N={n}: encode {te*1e3:.3f} ms, infer {ti*1e3:.3f} ms", flush=True)
