"""Per-call times of the bench entry `deblend_cutouts` (tools/field_cutouts.py::run_drop_in): DeblendField.deblend_field per
32768 galaxies, with the engine call timed separately.  python tools/probes/drop_in_calls.py [dtype] [full_warmup]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from debvader_amd import engine as E                                     # noqa: E402
from debvader_amd.deblend.field_deblender import DeblendField            # noqa: E402
from debvader_amd.model import model                                     # noqa: E402
from tools.field_cutouts import synthetic_field                          # noqa: E402

dtype = int(sys.argv[1]) if len(sys.argv) > 1 else 0
full_warm = len(sys.argv) > 2 and sys.argv[2] == "1"
field = np.asarray(synthetic_field(), np.float64)
field = field.reshape(field.shape[-3:])
scene = np.ascontiguousarray(np.tile(field, (8, 8, 1)))
F, cs, per_call, chunk = scene.shape[0], 59, 32768, 8192
starts = np.random.default_rng(0).integers(0, F - cs + 1, size=(6 * per_call, 2))
dist = (starts + cs // 2 - F // 2).astype(np.float64)
net, _, _, _ = model.create_model_vae((cs, cs, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=chunk, seed=0,
                                      dtype="bf16" if dtype else "float32")
db = DeblendField(net, scene[None])
eng = net._core.engine
t_eng = [0.0]
real = eng.infer_cutouts_keep


def timed(*a, **k):
    t = time.perf_counter()
    r = real(*a, **k)
    t_eng[0] += time.perf_counter() - t
    return r


eng.infer_cutouts_keep = timed
res = db.deblend_field(dist[:per_call if full_warm else 2 * chunk])
del res
for b in range(0, 6 * per_call, per_call):
    t_eng[0] = 0.0
    t0 = time.perf_counter()
    res = db.deblend_field(dist[b:b + per_call])
    del res
    dt = time.perf_counter() - t0
    print(f"call {b // per_call}: {dt * 1e3:7.1f} ms total, {t_eng[0] * 1e3:7.1f} ms in the engine call  ({per_call / dt:8.0f} stamps/s)  "
          f"pool {E.host_pool_stats()}", flush=True)
