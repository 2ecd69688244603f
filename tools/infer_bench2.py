import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.data import synthetic_stamps
N, MB = 16384, 2048
net, enc, dec, z = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=MB)
x, _ = synthetic_stamps(256, seed=1)
x = np.tile(x, (N // 256, 1, 1, 1))
x64 = x.astype(np.float64)
eng = net._core.engine
eng.infer(x[:4 * MB], seed=1)
for name, arr in (("f32", x), ("f64", x64), ("f32", x), ("f64", x64)):
    for want in (("loc", "scale"), ("mu",)):
        t0 = time.perf_counter(); r = eng.infer(arr, seed=1, want=want); dt = time.perf_counter() - t0
        print(name, want, f"{N/dt:.0f} stamps/s", flush=True)
        del r

out = {"loc": np.zeros((N, 59, 59, 6), np.float32), "scale": np.zeros((N, 59, 59, 6), np.float32)}
for i in range(3):
    t0 = time.perf_counter(); r = eng.infer(x, seed=1, out=out); dt = time.perf_counter() - t0
    print("f32 reused output arrays", f"{N/dt:.0f} stamps/s", flush=True)
for want in (("loc",), ("scale",), ("loc", "scale", "mu", "z")):
    t0 = time.perf_counter(); r = eng.infer(x, seed=1, want=want); dt = time.perf_counter() - t0
    print("f32", want, f"{N/dt:.0f} stamps/s", flush=True)
    del r
