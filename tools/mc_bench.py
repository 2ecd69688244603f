"""Epistemic Monte-Carlo path (dv_infer_mc): time per call for a few objects, 100 samples (the DeblendField use)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.deblend_cutout.deblender import deblend_epistemic
from debvader_amd.data import synthetic_stamps
net, _, _, _ = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=256)
x, _ = synthetic_stamps(256, seed=1)
for n in (1, 4, 16, 64, 256):
    deblend_epistemic(net, x[:n], n_samples=100)
    t0 = time.perf_counter()
    for _ in range(3):
        deblend_epistemic(net, x[:n], n_samples=100)
    dt = (time.perf_counter() - t0) / 3
    print(f"N={n:4d} x 100 samples: {dt*1e3:8.2f} ms per call = {n*100/dt:9.0f} decodes/s")
