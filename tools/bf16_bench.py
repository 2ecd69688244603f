"""Queued bf16 train steps on synthetic stamps (profiling target): python tools/bf16_bench.py [B] [steps] [dtype]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x, y = synthetic_stamps(B, seed=0)
eng = E.Engine(E.make_config(max_batch=B, dtype=dtype))
eng.optimizer_reset(1e-4)
eng.upload(0, x, y)
eng.train_steps(0, 0, B, 5, seed=1)
t0 = time.perf_counter()
out = eng.train_steps(0, 0, B, steps, seed=2)
dt = time.perf_counter() - t0
print(f"dtype {dtype} B {B}: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt:.0f} stamps/s  loss {out['loss']:.5g}")
eng.close()
