"""Independent second implementation of the conv-VAE in torch (CPU, fp64, autograd).

Test-only cross-check of oracle/vae_oracle.py: nothing here is imported by the
product.  Written directly from the Keras/TFP layer semantics, not from the oracle's code paths:
convolutions use torch.nn.functional with explicit asymmetric padding, gradients come from autograd.
"""
import math

import torch
import torch.nn.functional as F


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv_same(x, w, b, s):          # x NHWC, w HWIO
    xt = x.permute(0, 3, 1, 2)
    pt, pb = _same_pad(x.shape[1], w.shape[0], s)
    pl, pr = _same_pad(x.shape[2], w.shape[1], s)
    xt = F.pad(xt, (pl, pr, pt, pb))
    y = F.conv2d(xt, w.permute(3, 2, 0, 1), b, stride=s)
    return y.permute(0, 2, 3, 1)


def convt_same(x, k, b, s):         # k (kh,kw,cout,cin)
    xt = x.permute(0, 3, 1, 2)
    # torch weight for conv_transpose2d: (cin, cout, kh, kw)
    w = k.permute(3, 2, 0, 1)
    full = F.conv_transpose2d(xt, w, None, stride=s)      # size (in-1)*s + k
    Ho = x.shape[1] * s
    pt, _ = _same_pad(Ho, k.shape[0], s)
    pl, _ = _same_pad(x.shape[2] * s, k.shape[1], s)
    y = full[:, :, pt:pt + Ho, pl:pl + x.shape[2] * s] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def prelu(u, a):
    return torch.relu(u) - a * torch.relu(-u)


def fill_tril(v, n):
    xc = torch.cat([v[..., n:], torch.flip(v, dims=[-1])], dim=-1)
    return torch.tril(xc.reshape(v.shape[:-1] + (n, n)))


def net_loss(arch, p, x, y, eps, training=True):
    """Returns dict of torch scalars/tensors. p: dict name->torch tensor (requires_grad)."""
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = x.var(dim=(0, 1, 2), unbiased=False)
    else:
        mean, var = p["enc/bn/moving_mean"], p["enc/bn/moving_variance"]
    h = (x - mean) / torch.sqrt(var + arch.bn_eps) * p["enc/bn/gamma"] + p["enc/bn/beta"]
    for j in range(2 * len(arch.filters)):
        s = 2 if j % 2 == 1 else 1
        h = prelu(conv_same(h, p[f"enc/conv{j}/kernel"], p[f"enc/conv{j}/bias"], s), p[f"enc/prelu{j}/alpha"])
    B = x.shape[0]
    h = prelu(h.reshape(B, -1), p["enc/prelu_flat/alpha"])
    t = h @ p["enc/dense/kernel"] + p["enc/dense/bias"]
    d = arch.latent_dim
    mu = t[:, :d]
    L = fill_tril(t[:, d:], d)
    diag = F.softplus(torch.diagonal(L, dim1=-2, dim2=-1)) + arch.diag_shift
    L = L - torch.diag_embed(torch.diagonal(L, dim1=-2, dim2=-1)) + torch.diag_embed(diag)
    z = mu + torch.einsum("bij,bj->bi", L, eps)
    # single-sample MC KL via explicit log-densities (as TFP does): log q(z) - log p(z)
    sol = torch.linalg.solve_triangular(L, (z - mu).unsqueeze(-1), upper=False).squeeze(-1)
    logq = -0.5 * (sol ** 2).sum(1) - torch.log(diag).sum(1) - 0.5 * d * math.log(2 * math.pi)
    logp = -0.5 * (z ** 2).sum(1) - 0.5 * d * math.log(2 * math.pi)
    kl = logq - logp
    h = prelu(z, p["dec/prelu_in/alpha"])
    h = prelu(h @ p["dec/dense0/kernel"] + p["dec/dense0/bias"], p["dec/prelu_h/alpha"])
    h = prelu(h @ p["dec/dense1/kernel"] + p["dec/dense1/bias"], p["dec/prelu_r/alpha"])
    h = h.reshape(B, arch.w0, arch.w0, arch.filters[-1])
    for j in range(2 * len(arch.filters)):
        s = 2 if j % 2 == 0 else 1
        h = prelu(convt_same(h, p[f"dec/convt{j}/kernel"], p[f"dec/convt{j}/bias"], s), p[f"dec/prelut{j}/alpha"])
    tt = torch.relu(conv_same(h, p["dec/head/kernel"], p["dec/head/bias"], 1))
    c0 = arch.crop[0]
    H = arch.input_shape[0]
    tt = tt[:, c0:c0 + H, c0:c0 + H]
    loc, scale = tt[..., :arch.nb], arch.sigma_floor + tt[..., arch.nb:]
    dist = torch.distributions.Normal(loc, scale)
    nll = -dist.log_prob(y)
    nll_mean = nll.mean()
    kl_reg = arch.kl_multiplicity * arch.kl_weight * kl.mean() / B
    return dict(loss=nll_mean + kl_reg, nll_mean=nll_mean, kl_reg=kl_reg, t=t, z=z, loc=loc, scale=scale, kl=kl)
