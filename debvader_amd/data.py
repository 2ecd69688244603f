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
    sed, sig_band = _SED[:nb], _NOISE[:nb]
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
