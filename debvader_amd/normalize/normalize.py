"""Stamp normalisation used around the network by `deblend(..., normalise=True)`.

The reference (`/root/reference/src/debvader/normalize/normalize.py:3-7`) squashes fluxes with tanh(arcsinh(x)) and
undoes it with sinh(arctanh(y)).  These host versions serve callers that normalise their own arrays; inside
`deblend` the same two maps run as GPU kernels (`dv_model_set_normalise`, csrc/pointwise.hip), in the closed forms
x / sqrt(1 + x^2) and y / sqrt(1 - y^2).
"""
import numpy as np

__all__ = ["normalize_non_linear", "denormalize_non_linear"]


def _as_float(a):
    a = np.asarray(a)
    return a if np.issubdtype(a.dtype, np.floating) else a.astype(np.float64)


def normalize_non_linear(images):
    """Flux -> (-1, 1): tanh of the inverse hyperbolic sine, elementwise."""
    flux = _as_float(images)
    return np.tanh(np.arcsinh(flux))


def denormalize_non_linear(images_normed):
    """Inverse of `normalize_non_linear`; |values| >= 1 map to +-inf / nan exactly as numpy's arctanh does."""
    squashed = _as_float(images_normed)
    return np.sinh(np.arctanh(squashed))
