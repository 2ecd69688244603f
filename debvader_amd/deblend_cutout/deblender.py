"""Batched inference entry point (reference: src/debvader/deblend_cutout/deblender.py:6-24)."""
import numpy as np

from debvader_amd.distributions import Normal
from debvader_amd.normalize.normalize import denormalize_non_linear, normalize_non_linear


def deblend(net, images, normalise=False):
    """Deblend stamps with the network.

    parameters:
        net: network returned by create_model_vae / load_deblender
        images: array (N, size, size, bands), any float dtype (cast to float32 as the reference does)
        normalise: apply tanh(arcsinh(.)) to the inputs and undo it on the predicted mean.
            The reference's normalise=True branch cannot run (it applies numpy arctanh to a
            distribution object, deblender.py:20-24); the evident intent is implemented instead.
    returns (mean ndarray (N,size,size,bands), distribution)
    """
    images = np.asarray(images)
    if normalise:
        images = normalize_non_linear(images)
    out = net(images)     # one stochastic forward pass, BN in inference mode; the float32 cast happens in the engine
    if normalise:
        mean = denormalize_non_linear(np.clip(out.mean().numpy(), -1 + 1e-7, 1 - 1e-7))
        return mean, Normal(mean, out.stddev().numpy())
    return out.mean().numpy(), out


def deblend_epistemic(net, images, n_samples=100, normalise=False):
    """Epistemic-uncertainty estimate of the reference's field deblender (deblend/field_deblender.py:303-313:
    `np.std(deblend(net, [stamp] * 100)[0], axis=0)` in a Python loop over objects) as one engine call:
    every stamp is encoded once and decoded `n_samples` times with fresh latent samples on the GPU.

    returns (mean over samples of the predicted mean, std over samples), each (N, size, size, bands)
    """
    images = np.asarray(images)
    if normalise:
        images = normalize_non_linear(images)
    mean, std = net._core.engine.infer_mc(images.astype(np.float32), n_samples, seed=net._core.next_seed())
    if normalise:
        mean = denormalize_non_linear(np.clip(mean, -1 + 1e-7, 1 - 1e-7))
    return mean, std
