"""Third probe (CPU only): is the distance between a float32 evaluation and float64 on the 128-px /
64-stamp case carried by GATES alone?  float64 forward and backward, but every PReLU / relu gate (u > 0) takes the state it
has in the float32 forward - nothing else of the float32 evaluation is used."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import vae_oracle as vo               # noqa: E402
from tests import oracle_jobs as oj               # noqa: E402


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
    n2 = 2 * len(arch.filters)
    gates = [f"enc_u{j}" for j in range(n2)] + ["enc_flat_u", "dec_z", "dec_u_h", "dec_u_r"] + [f"dec_u{j}" for j in range(n2)] + ["head_pre"]
    cm = dict(c64)
    nflip = 0
    for n in gates:
        a, b = c64[n], c32[n]
        m = (a > 0) != (b > 0)
        if m.any():
            a = a.copy()
            a[m] = np.where(b[m] > 0, 1e-300, -1e-300)     # the float32 gate state, no other change
            cm[n] = a
            nflip += int(m.sum())
    gm = vo.backward(arch, p, cm, y.astype(np.float64))
    print(f"{nflip} gates differ between the float32 and the float64 forward")
    rows = sorted(((rel(g32[k], g64[k]), rel(g32[k], gm[k]), rel(gm[k], g64[k]), k) for k in g64), reverse=True)
    print(f"{'tensor':26s} {'f32 vs f64':>11s} {'f32 vs f64 with f32 gates':>26s} {'gates alone':>12s}")
    for e0, e1, e2, k in rows[:24]:
        print(f"{k:26s} {e0:11.3e} {e1:26.3e} {e2:12.3e}")
    print("largest remaining distance (float32 vs gate-matched float64):", max(r[1] for r in rows))


if __name__ == "__main__":
    main()
