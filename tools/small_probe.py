"""One-stamp and eight-stamp inference latency (encode / infer per call) - and, run as
`rocprofv3 --kernel-trace -d gpurun_out/small_prof -- python3 tools/small_probe.py 10`, the kernel trace DESIGN.md 7a
reads (27 kernels of a one-stamp forward).  The script leaves its Engine and Context to the interpreter's exit on
purpose: debvader_amd.engine closes them in an atexit hook (models first, then the context), which is what this probe
checks under the profiler (SIGSEGV inside exit() when they were left to __del__)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
eng = E.Engine(E.make_config(max_batch=32))
eng.init(seed=1)
x, _ = synthetic_stamps(8, seed=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
    print(f"N={n}: encode {te*1e3:.3f} ms, infer {ti*1e3:.3f} ms", flush=True)
print("leaving without close(): the atexit hook of debvader_amd.engine tears down", flush=True)
