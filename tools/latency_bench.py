"""deblend() latency for small numbers of stamps (the per-field use of the reference's DeblendField)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.deblend_cutout.deblender import deblend
from debvader_amd.data import synthetic_stamps
net, _, _, _ = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=256)
x, _ = synthetic_stamps(256, seed=1)
for n in (1, 8, 32, 64, 128, 256):
    for _ in range(3):
        deblend(net, x[:n])
    t0 = time.perf_counter()
    for _ in range(20):
        deblend(net, x[:n])
    dt = (time.perf_counter() - t0) / 20
    print(f"N={n:4d}: {dt*1e3:7.3f} ms per call = {n/dt:8.0f} stamps/s")
