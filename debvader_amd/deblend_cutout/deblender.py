"""Batched inference entry point (reference: src/debvader/deblend_cutout/deblender.py:6-24)."""
import numpy as np


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
    if not normalise:
        out = net(images)  # one stochastic forward pass, BN in inference mode; the float32 cast happens in the engine
        return out.mean().numpy(), out
    eng = net._core.engine
    eng.set_normalise(True)                        # both transforms run on the GPU around the forward pass
    try:
        out = net(images)
    finally:
        eng.set_normalise(False)
    return out.mean().numpy(), out                 # mean in flux units, stddev in normalised units

def deblend_epistemic(net, images, n_samples=100, normalise=False):
    """Epistemic-uncertainty estimate of the reference's field deblender (deblend/field_deblender.py:303-313:
    `np.std(deblend(net, [stamp] * 100)[0], axis=0)` in a Python loop over objects) as one engine call:
    every stamp is encoded once and decoded `n_samples` times with fresh latent samples on the GPU.

    returns (mean over samples of the predicted mean, std over samples), each (N, size, size, bands)
    """
    images = np.asarray(images)
    eng = net._core.engine
    eng.set_normalise(bool(normalise))
    try:
        mean, std = eng.infer_mc(images.astype(np.float32), n_samples, seed=net._core.next_seed())
    finally:
        eng.set_normalise(False)
    return mean, std
