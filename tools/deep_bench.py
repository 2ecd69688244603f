"""Train-step throughput of BASELINE configs[3]: 128x128x6 stamps, six-level encoder / decoder
(filters [32,64,128,256,512,512], SURVEY 8(d): 3.53 GFLOP forward per stamp), 64 stamps per GPU.  GPU only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E

B = int(os.environ.get("DB_BATCH", "64"))
cfg = E.make_config(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512),
                    kernels=(3,) * 6, max_batch=B)
eng = E.Engine(cfg)
eng.init(seed=1)
eng.optimizer_reset(1e-4)
rng = np.random.default_rng(0)
x = rng.normal(0.05, 0.3, size=(2 * B, 128, 128, 6)).astype(np.float32)
eng.upload(0, x, x)
eng.train_steps(0, 0, B, 5, seed=1)
t0 = time.perf_counter()
K = 40
out = eng.train_steps(0, 0, B, K, seed=2)
dt = (time.perf_counter() - t0) / K
fwd_flops = 3.53e9
print(f"128x128x6, 6 levels, batch {B}: {dt*1e3:.2f} ms/step = {B/dt:.0f} stamps/s = {3*fwd_flops*B/dt/1e12:.1f} TFLOP/s; loss {out['loss']:.4g}")
