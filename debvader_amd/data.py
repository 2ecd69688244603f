"""Synthetic 6-band galaxy stamps: Gaussian-blob scenes with the value
range of the reference's DC2 samples (src/debvader/data/dc2_imgs/imgs_dc2.npy: min -0.97, max 14.2,
mean 0.047).  Used by bench.py, smoke() and the tests; there is no network for real datasets.

label  = one centred elliptical Gaussian blob x band SED
input  = label + 0..3 neighbour blobs at random offsets + per-band Gaussian noise
"""
from __future__ import annotations

import math

import numpy as np

_SED = np.array([0.16, 0.24, 0.40, 0.62, 0.85, 1.0, 1.0, 1.0])
_NOISE = np.array([0.02, 0.03, 0.05, 0.08, 0.10, 0.11, 0.11, 0.11])


def synthetic_stamps(n: int, seed: int = 0, size: int = 59, nb: int = 6, dtype=np.float32):
    """Returns (blended inputs, isolated labels), each (n, size, size, nb)."""
    rng = np.random.default_rng(seed)
    # (more than six bands: the measured six-band SED / noise rows repeat)
    sed, sig_band = np.resize(_SED, nb), np.resize(_NOISE, nb)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    c0 = (size - 1) / 2.0

    def blob(cx, cy):
        s = rng.uniform(1.5, 4.0)
        q = rng.uniform(0.5, 1.0)
        th = rng.uniform(0, np.pi)
        peak = math.exp(rng.uniform(math.log(0.5), math.log(15.0)))
        dx, dy = xx - cx, yy - cy
        u = dx * math.cos(th) + dy * math.sin(th)
        v = -dx * math.sin(th) + dy * math.cos(th)
        return peak * np.exp(-0.5 * (u * u / (s * s) + v * v / (s * s * q * q)))

    X = np.empty((n, size, size, nb), dtype=dtype)
    Y = np.empty((n, size, size, nb), dtype=dtype)
    for i in range(n):
        lab = blob(c0, c0)
        img = lab.copy()
        for _ in range(rng.integers(0, 4)):
            img += blob(c0 + rng.uniform(-20, 20), c0 + rng.uniform(-20, 20))
        noise = rng.normal(0.0, 1.0, size=(size, size, nb)) * sig_band
        Y[i] = (lab[..., None] * sed).astype(dtype)
        X[i] = (img[..., None] * sed + noise).astype(dtype)
    return X, Y


def dc2_sample_stamps(path=None):
    """The reference's real sample stamps this repository carries: the first four 59x59x6 DC2 stamps of
    src/debvader/data/dc2_imgs/imgs_dc2.npy, float32, kept as the inputs of the golden fixture tests/golden/dc2_b4.npz.
    Returns (x, y) or None when the fixture is not there (an installed package without the test tree)."""
    import os

    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dc2_b4.npz")
    if not os.path.exists(path):
        return None
    with np.load(path) as d:
        return np.asarray(d["x"], np.float32), np.asarray(d["y"], np.float32)


def bench_stamps(n: int, seed: int = 0, real_fraction: float = 0.125):
    """real DC2 stamps tiled + the config-1 generator to fill the batch.  Every 1/real_fraction-th
    row is one of the four real stamps (cycled), the others come from synthetic_stamps(seed).  Returns (x, y, description);
    the kernels' run time does not depend on the values, the description says what the rows were."""
    x, y = synthetic_stamps(n, seed=seed)
    real = dc2_sample_stamps()
    if real is None or real[0].shape[1:] != x.shape[1:]:
        return x, y, "synthetic"
    step = max(1, int(round(1.0 / real_fraction)))
    rows = np.arange(0, n, step)
    x[rows] = real[0][np.arange(rows.size) % real[0].shape[0]]
    y[rows] = real[1][np.arange(rows.size) % real[1].shape[0]]
    return x, y, (f"synthetic Gaussian-blob stamps (SURVEY 8(d) config 1 generator) with the 4 real DC2 stamps of "
                  f"tests/golden/dc2_b4.npz (from the reference's imgs_dc2.npy) tiled into every {step}th row; random-init weights")
