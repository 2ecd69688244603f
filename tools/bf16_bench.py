"""Queued bf16 train steps on synthetic stamps (profiling target): python tools/bf16_bench.py [B] [steps] [dtype]"""
import sys
import time

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the MEASUREMENT switches (DV_EXP_*, DV_TIME_ENQUEUE: work left out, wrong results) exist in the development build only;
# when one is set this tool drives that build (the product library does not read them)
if any(k.startswith("DV_EXP_") or k == "DV_TIME_ENQUEUE" for k in os.environ) and not os.environ.get("DEBVADER_AMD_LIB"):
    os.environ["DEBVADER_AMD_LIB"] = os.path.join(ROOT, "debvader_amd", "lib", "libdebvader_hip_debug.so")
    print("[bf16_bench] measurement switch set: using the development build of the engine", file=sys.stderr)
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
