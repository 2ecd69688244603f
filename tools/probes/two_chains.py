"""Two independent half-batch training chains on one GPU (a probe for "two stamp-halves as two streams"): two Engine
objects with their own Context (stream set), B stamps each, queued from two threads; prints the aggregate stamps/s beside
one engine at 2B.  python tools/probes/two_chains.py [B per chain] [steps] [dtype]"""
import os
import sys
import threading
import time

sys.path.insert(0, ".")
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def make(ctx, b):
    x, y = synthetic_stamps(b, seed=0)
    eng = E.Engine(E.make_config(max_batch=b, dtype=dtype), ctx=ctx)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    eng.train_steps(0, 0, b, 5, seed=1)
    return eng


ctx0 = E.default_context()
one = make(ctx0, 2 * B)
t0 = time.perf_counter()
one.train_steps(0, 0, 2 * B, steps, seed=2)
t1 = time.perf_counter() - t0
print(f"one engine, {2 * B} stamps per step: {t1 / steps * 1e3:.3f} ms/step, {2 * B * steps / t1:.0f} stamps/s")
one.close()
ctxs = [ctx0, E.Context()]
engs = [make(c, B) for c in ctxs]
bar = threading.Barrier(3)


def run(e):
    bar.wait()
    e.train_steps(0, 0, B, steps, seed=3)
    bar.wait()


th = [threading.Thread(target=run, args=(e,)) for e in engs]
for t in th:
    t.start()
bar.wait()
t0 = time.perf_counter()
bar.wait()
t2 = time.perf_counter() - t0
for t in th:
    t.join()
print(f"two engines, {B} stamps per step each, concurrently: {t2 / steps * 1e3:.3f} ms per pair of steps, "
      f"{2 * B * steps / t2:.0f} stamps/s aggregate")
for e in engs:
    e.close()
