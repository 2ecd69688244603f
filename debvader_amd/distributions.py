"""Distribution-like return values of net(x), decoder(z) and z(x).

The reference returns tfp.distributions objects (model.py:154-159, 211-214) and its callers use
.mean(), .stddev(), .sample(n), .log_prob(y) and .numpy() on the results (deblender.py:24,
field_deblender.py:368-370, notebooks/behavior_of_latent_space.ipynb:186-200,294-296).  These thin
wrappers hold the arrays the engine produced; they do not compute the network.
"""
from __future__ import annotations

import math

import numpy as np


class Tensor(np.ndarray):
    """ndarray with the .numpy() accessor TF eager tensors have."""

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def numpy(self):
        return np.asarray(self)


def _shape(n):
    if n is None or n == ():
        return ()
    if isinstance(n, (int, np.integer)):
        return (int(n),)
    return tuple(int(v) for v in n)


class Normal:
    """tfd.Normal(loc, scale) over (N,H,W,C) (model.py:155-157); not wrapped in Independent."""

    def __init__(self, loc, scale, seed=None):
        self.loc = np.asarray(loc, dtype=np.float32)
        self.scale = np.asarray(scale, dtype=np.float32)
        self._rng = np.random.default_rng(seed)

    def mean(self):
        return Tensor(self.loc)

    def stddev(self):
        return Tensor(self.scale)

    def variance(self):
        return Tensor(self.scale * self.scale)

    def sample(self, sample_shape=(), seed=None):
        rng = self._rng if seed is None else np.random.default_rng(seed)
        shp = _shape(sample_shape) + self.loc.shape
        return Tensor(self.loc + self.scale * rng.standard_normal(shp, dtype=np.float32))

    def log_prob(self, value):
        # TFP form: -0.5*((x/s) - (m/s))^2 - log(s) - 0.5*log(2*pi)
        value = np.asarray(value, dtype=np.float32)
        z = (value - self.loc) / self.scale
        return Tensor(-0.5 * z * z - np.log(self.scale) - np.float32(0.5 * math.log(2.0 * math.pi)))

    def numpy(self):  # convert_to_tensor_fn = sample (model.py:158)
        return self.sample().numpy()

    @property
    def batch_shape(self):
        return self.loc.shape


def fill_triangular(v):
    """tfp.math.fill_triangular, lower (op order of deblender_to_onnx.ipynb:160-187)."""
    v = np.asarray(v)
    m = v.shape[-1]
    n = int((math.isqrt(1 + 8 * m) - 1) // 2)
    xc = np.concatenate([v[..., n:], v[..., ::-1]], axis=-1)
    return np.tril(xc.reshape(v.shape[:-1] + (n, n)))


class MultivariateNormalTriL:
    """Posterior q(z|x) built from the encoder output t (model.py:43-58, 211-214)."""

    def __init__(self, t, latent_dim, diag_shift=1e-5, seed=None):
        t = np.asarray(t, dtype=np.float32)
        d = latent_dim
        self.loc = t[..., :d]
        L = fill_triangular(t[..., d:]).astype(np.float32)
        i = np.arange(d)
        L[..., i, i] = np.logaddexp(0.0, L[..., i, i]) + np.float32(diag_shift)
        self.scale_tril = L
        self._rng = np.random.default_rng(seed)

    def mean(self):
        return Tensor(self.loc)

    def stddev(self):
        return Tensor(np.sqrt((self.scale_tril ** 2).sum(-1)))

    def covariance(self):
        return Tensor(self.scale_tril @ np.swapaxes(self.scale_tril, -1, -2))

    def sample(self, sample_shape=(), seed=None):
        rng = self._rng if seed is None else np.random.default_rng(seed)
        shp = _shape(sample_shape) + self.loc.shape
        eps = rng.standard_normal(shp, dtype=np.float32)
        return Tensor(self.loc + np.einsum("...ij,...j->...i", self.scale_tril, eps))

    def numpy(self):
        return self.sample().numpy()
