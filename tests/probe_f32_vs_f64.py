"""Where does a float32 evaluation of the 128-px / 64-stamp step leave the float64 oracle? (TEST
INFRASTRUCTURE, CPU only: python tests/probe_f32_vs_f64.py [out.npz]).  Evaluates the case of
tests/test_gpu_0_fullsize_oracle.py::test_128px_six_level_arch_at_its_per_gpu_batch_of_64 in float64 and in numpy float32,
and looks for DISCRETE events between the two: PReLU / relu gates whose sign differs, per layer, and what the gradient
differences of the decoder-trunk tensors look like (one stamp? one unit? spread?)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import vae_oracle as vo               # noqa: E402
from tests import oracle_jobs as oj               # noqa: E402


def main():
    arch = oj.make_arch(oj.DEEP)
    B, seed = 64, 21
    p, x, y, eps = oj.f32_case_inputs(arch, B, seed, None, 0.3)
    t0 = time.time()
    c64 = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=True)
    g64 = vo.backward(arch, p, c64, y.astype(np.float64))
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    c32 = vo.forward(arch, p32, x, eps, training=True)
    g32 = vo.backward(arch, p32, c32, y)
    print(f"evaluated in {time.time() - t0:.0f} s")
    rel = {k: float(np.abs(g32[k] - g64[k]).max() / np.abs(g64[k]).max()) for k in g64}
    for k in sorted(rel, key=lambda k: -rel[k])[:14]:
        print(f"  {k:28s} {rel[k]:.3e}")
    # gates whose sign differs between the two evaluations
    n2 = 2 * len(arch.filters)
    names = [f"enc_u{j}" for j in range(n2)] + ["enc_flat_u", "dec_z", "dec_u_h", "dec_u_r"] + [f"dec_u{j}" for j in range(n2)] + ["head_pre"]
    flips = {}
    for n in names:
        if n not in c64:
            continue
        a, b = c64[n], c32[n]
        m = (a > 0) != (b > 0)
        if m.any():
            idx = np.argwhere(m)
            flips[n] = idx
            print(f"  gate flips in {n}: {len(idx)} of {a.size}; first {idx[:4].tolist()}; |u64| there <= {np.abs(a[m]).max():.2e}")
    # shape of the gradient difference of the tensors the margins file lists
    for k in ("dec/prelu_in/alpha", "dec/dense0/kernel", "dec/dense0/bias", "dec/prelu_h/alpha", "dec/dense1/bias"):
        d = (g32[k] - g64[k]).astype(np.float64)
        flat = np.abs(d).ravel()
        o = np.argsort(-flat)[:5]
        print(f"  {k}: |diff| max {flat[o[0]]:.3e} at {np.unravel_index(o[0], d.shape)}, next {flat[o[1:]].tolist()}, "
              f"median {np.median(flat):.3e}, max|g64| {np.abs(g64[k]).max():.3e}")
    if len(sys.argv) > 1:
        np.savez(sys.argv[1], **{k.replace("/", "__"): (g32[k] - g64[k]) for k in g64})
    return c64, c32, g64, g32, flips


if __name__ == "__main__":
    main()
