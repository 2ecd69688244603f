"""Reconstruction loss and image MSE (reference: src/debvader/training/metrics.py:4-26)."""
import numpy as np


def mse(img1, img2):
    """Mean squared error between two images (metrics.py:4-12)."""
    return np.mean(np.square(np.asarray(img1) - np.asarray(img2)))


def vae_loss(ground_truth, predicted_distribution):
    """Per-pixel negative log-likelihood of the ground truth under the predicted Normal
    (metrics.py:16-26).  Inside fit() the engine evaluates exactly this on the GPU, fused with
    the relu/crop head; this host version serves callers that hold a distribution object."""
    return -predicted_distribution.log_prob(ground_truth)
