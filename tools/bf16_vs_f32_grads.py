"""Per-tensor comparison of the bf16 and the fp32 engine's gradients for one step from the same state, and of their loss
curves over a few Adam steps (diagnosis tool behind tests/test_gpu_bf16.py::test_bf16_training_tracks_fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps

B = 64
x, y = synthetic_stamps(2 * B, seed=21)
engs = []
for dtype in (0, 1):
    eng = E.Engine(E.make_config(max_batch=B, dtype=dtype))
    eng.init(seed=5)
    hb = eng.get_param("dec/head/bias")
    hb[6:] += 0.3
    eng.set_param("dec/head/bias", hb)
    eng.optimizer_reset(float(os.environ.get("LR", "1e-4")))
    eng.upload(0, x, y)
    engs.append(eng)
f, b = engs
of = f.grad_step(0, first=0, B=B, seed=100)
ob = b.grad_step(0, first=0, B=B, seed=100)
print("loss", of["loss"], ob["loss"])
for name, shape, tr in f.specs:
    if not tr:
        continue
    gf = f.get_grad(name).astype(np.float64).ravel()
    gb = b.get_grad(name).astype(np.float64).ravel()
    nf, nb_ = np.linalg.norm(gf), np.linalg.norm(gb)
    cos = gf.dot(gb) / (nf * nb_ + 1e-300)
    print(f"{name:28s} |g32| {nf:10.3e} ratio {nb_ / (nf + 1e-300):6.3f} cos {cos:7.4f} relmax {np.abs(gf - gb).max() / (np.abs(gf).max() + 1e-300):7.4f}")
steps = int(os.environ.get("STEPS", "40"))
lf, lb = [], []
for s in range(steps):
    lf.append(f.train_step(0, first=(s % 2) * B, B=B, seed=200 + s)["loss"])
    lb.append(b.train_step(0, first=(s % 2) * B, B=B, seed=200 + s)["loss"])
print("fp32", np.round(lf[::4], 4))
print("bf16", np.round(lb[::4], 4))
