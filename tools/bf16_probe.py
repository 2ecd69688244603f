"""Development probe for the bf16 engine: deviations from the bf16-storage oracle (kernel check) and from the fp64
oracle (precision of the format), then a timing of queued train steps.  Usage: python tools/bf16_probe.py [quick]"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae_oracle as vo
from oracle import vae_oracle_bf16 as vb
from debvader_amd import engine as E


def relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def case(arch, B, seed, data=None):
    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    H, W, C = arch.input_shape
    if data is None:
        x = rng.normal(0, 0.4, size=(B, H, W, C)).astype(np.float32)
        y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C))).astype(np.float32)
    else:
        x, y = data
    eps = rng.normal(size=(B, arch.latent_dim)).astype(np.float32)
    if "floor" not in sys.argv:
        # sigma well above its 1e-4 floor: at the floor 1/sigma^2 = 1e8 turns a one-ulp bf16 difference of the mean
        # into an O(1) change of the gradient, and no two implementations agree
        p["dec/head/bias"][arch.nb:] += 0.3
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return p, x, y, eps


def probe(arch, B, seed, data=None, label=""):
    p, x, y, eps = case(arch, B, seed, data)
    cfg = E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels), max_batch=B, dtype=1)
    eng = E.Engine(cfg)
    eng.set_params(p)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    eng.keep_outputs(True)
    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
    fused = ((B + 15) // 16 * 16) % 64 == 0
    cb = vb.forward(arch, p, x64, e64, training=True)
    rb = vo.losses(arch, cb, y64)
    gb = vb.backward(arch, p, cb, y64, fused=fused)
    c = vo.forward(arch, p, x64, e64, training=True)
    r = vo.losses(arch, c, y64)
    g = vo.backward(arch, p, c, y64)
    out = eng.grad_step(0, first=0, B=B, eps=eps)
    H, W, C = arch.input_shape
    d = arch.latent_dim
    print(f"== {label} B={B} fused={fused}")
    for k, shape in (("t", (B, arch.params_size)), ("z", (B, d)), ("kl", (B,)), ("loc", (B, H, W, C)),
                     ("scale", (B, H, W, C)), ("head_pre", (B, arch.dec_out, arch.dec_out, 2 * C))):
        v = eng.activation(k, shape)
        print(f"  act {k:9s} vs bf16-oracle {relmax(v, cb[k]):.2e}   vs fp64 {relmax(v, c[k]):.2e}")
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        print(f"  {k:9s} gpu {out[k]:.6g}  bf16-oracle {rb[k]:.6g} ({abs(out[k]-rb[k])/abs(rb[k]):.2e})  fp64 {r[k]:.6g} ({abs(out[k]-r[k])/abs(r[k]):.2e})")
    wb = w64 = ("", 0.0)
    for name in g:
        gg = eng.get_grad(name)
        eb, e64_ = relmax(gg, gb[name]), relmax(gg, g[name])
        if eb > 2e-3 or "-v" in sys.argv:
            print(f"    grad {name:28s} vs bf16-oracle {eb:.2e}  vs fp64 {e64_:.2e}   (bf16-oracle vs fp64 {relmax(gb[name], g[name]):.2e})")
        if eb > wb[1]:
            wb = (name, eb)
        if e64_ > w64[1]:
            w64 = (name, e64_)
    print(f"  worst grad vs bf16-oracle {wb}, vs fp64 {w64}")
    eng.close()


def timing(B=256, steps=50):
    from debvader_amd.data import synthetic_stamps
    x, y = synthetic_stamps(B, seed=0)
    for dtype in (1, 0):
        cfg = E.make_config(max_batch=B, dtype=dtype)
        eng = E.Engine(cfg)
        eng.optimizer_reset(1e-4)
        eng.upload(0, x, y)
        eng.train_steps(0, 0, B, 10, seed=1)
        t0 = time.perf_counter()
        out = eng.train_steps(0, 0, B, steps, seed=2)
        dt = time.perf_counter() - t0
        print(f"dtype {dtype}: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt:.0f} stamps/s  loss {out['loss']:.5g}")
        eng.prof_enable(True)
        eng.train_steps(0, 0, B, 5, seed=3)
        for k, nm in enumerate(("conv", "wgrad", "other")):
            n, ms = eng.prof_read(k)
            print(f"   class {nm}: {n / 5:.0f} launches/step, {ms / 5:.3f} ms/step")
        eng.close()


if __name__ == "__main__":
    toy = vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3))
    probe(toy, 5, 0, label="toy")
    probe(toy, 64, 1, label="toy")
    if "quick" not in sys.argv:
        from debvader_amd.data import synthetic_stamps
        x, y = synthetic_stamps(4, seed=5)
        probe(vo.Arch(), 4, 2, data=(x, y), label="full")
        x, y = synthetic_stamps(64, seed=6)
        probe(vo.Arch(), 64, 3, data=(x, y), label="full")
    timing()
