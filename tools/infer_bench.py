"""deblend() throughput (BASELINE configs[4] shape: batches of 8192 cutouts of a 259x259x6 scene), 1 GPU.
Reports stamps/s including the host<->device copies the reference-compatible API implies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.deblend_cutout.deblender import deblend
from debvader_amd.data import synthetic_stamps

N = int(os.environ.get("IB_N", "16384"))
MB = int(os.environ.get("IB_BATCH", "2048"))
net, enc, dec, z = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=MB)
x, _ = synthetic_stamps(256, seed=1)
x = np.tile(x, (N // 256, 1, 1, 1))
deblend(net, x[:MB])
deblend(net, x[:4 * MB])      # sizes the pinned staging buffers of the pipeline
t0 = time.perf_counter()
mean, dist = deblend(net, x)
dt = time.perf_counter() - t0
print(f"deblend(): {N} stamps in {dt*1e3:.1f} ms = {N/dt:.0f} stamps/s (engine chunk {MB}, incl. H2D/D2H of 250 KB/stamp)")
eng = net._core.engine
t0 = time.perf_counter()
r = eng.infer(x, seed=1, want=("mu",))
dt = time.perf_counter() - t0
print(f"infer() latent means only: {N/dt:.0f} stamps/s (H2D of 83.5 KB/stamp, no image D2H)")
x64 = x.astype(np.float64)
del mean, dist, r          # freeing 2.7 GB of results inside the timed region would be charged to it
t0 = time.perf_counter()
mean, dist = deblend(net, x64)
dt = time.perf_counter() - t0
print(f"deblend() on float64 stamps (the reference's input dtype): {N/dt:.0f} stamps/s")
