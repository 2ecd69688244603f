import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.data import synthetic_stamps
N, MB = 16384, 2048
net, enc, dec, z = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=MB)
x, _ = synthetic_stamps(256, seed=1)
x = np.tile(x, (N // 256, 1, 1, 1))
eng = net._core.engine
eng.infer(x[:4 * MB], seed=1)
print("---- timed", file=sys.stderr, flush=True)
t0 = time.perf_counter(); r = eng.infer(x, seed=1); dt = time.perf_counter() - t0
print(f"{N/dt:.0f} stamps/s")
