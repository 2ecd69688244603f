"""Second probe (CPU only): which HALF of a float32 evaluation carries the distance from float64 on
the decoder-trunk gradients of the 128-px / 64-stamp case - the forward values or the backward arithmetic?
  A: float32 forward, backward in float64 on those values     B: float64 forward, backward in float32 on those values"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import vae_oracle as vo               # noqa: E402
from tests import oracle_jobs as oj               # noqa: E402

KEYS = ("dec/prelu_in/alpha", "dec/dense0/kernel", "dec/dense0/bias", "dec/prelu_h/alpha", "dec/dense1/kernel", "dec/convt2/kernel",
        "dec/convt0/kernel", "dec/head/kernel", "enc/dense/kernel", "enc/prelu0/alpha")


def cast(c, dt):
    return {k: (v.astype(dt) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in c.items()}


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


def main():
    arch = oj.make_arch(oj.DEEP)
    p, x, y, eps = oj.f32_case_inputs(arch, 64, 21, None, 0.3)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    c64 = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=True)
    g64 = vo.backward(arch, p, c64, y.astype(np.float64))
    c32 = vo.forward(arch, p32, x, eps, training=True)
    g32 = vo.backward(arch, p32, c32, y)
    gA = vo.backward(arch, p, cast(c32, np.float64), y.astype(np.float64))
    gB = vo.backward(arch, p32, cast(c64, np.float32), y)
    print(f"{'tensor':24s} {'f32 all':>10s} {'A f32 fwd':>10s} {'B f32 bwd':>10s}")
    for k in KEYS:
        print(f"{k:24s} {rel(g32[k], g64[k]):10.3e} {rel(gA[k], g64[k]):10.3e} {rel(gB[k], g64[k]):10.3e}")
    # which forward tensor? replace ONE group of float32 forward values at a time in the float64 cache
    groups = {"head (head_pre, loc, scale)": ("head_pre", "loc", "scale", "head_in"),
              "decoder stack": tuple(k for k in c64 if k.startswith("dec_u") and k[5:].isdigit() or k.startswith("dec_in")),
              "trunk (z, a_in, u_h, a_h, u_r)": ("dec_z", "dec_a_in", "dec_u_h", "dec_a_h", "dec_u_r", "z", "L", "Lraw", "t", "mu")}
    for name, keys in groups.items():
        cm = dict(c64)
        for k in keys:
            if k in c32:
                cm[k] = c32[k].astype(np.float64)
        gm = vo.backward(arch, p, cm, y.astype(np.float64))
        print(f"float64 everywhere, float32 values for {name}: " + ", ".join(f"{k.split('/')[1]} {rel(gm[k], g64[k]):.2e}" for k in KEYS[:6]))


if __name__ == "__main__":
    main()
