"""deblend() latency for small numbers of stamps (the per-object use of the reference's DeblendField,
deblend/field_deblender.py:265-274): per call, host copies included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.deblend_cutout.deblender import deblend
from debvader_amd.data import synthetic_stamps
net, _, _, _ = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=256)
x, _ = synthetic_stamps(256, seed=1)
eng = net._core.engine


def timed(n, reps=50):
    for _ in range(5):
        deblend(net, x[:n])
    t0 = time.perf_counter()
    for _ in range(reps):
        deblend(net, x[:n])
    return (time.perf_counter() - t0) / reps


for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    a = timed(n)
    print(f"N={n:4d}: {a*1e3:7.3f} ms per call | {n/a:8.0f} stamps/s", flush=True)
