"""CPU oracle for the debvader conv-VAE hot path (TEST INFRASTRUCTURE, not product).

PARITY UNPINNED: the reference's arithmetic lives in tensorflow==2.13.0 /
tensorflow-probability==0.21.0 (requirements.txt:9-10), which are absent from
/root/reference and from this image, and the reference's own tests hold no
golden vector for this path (tests/test_extraction.py is the only test).  This
file is a numpy restatement of the reference's algorithm, following the
reference source lines cited on every function below, plus the published
semantics of the pinned Keras/TFP layers.  It is cross-checked in tests/ against
an independent torch-CPU autograd implementation and fp64 finite differences.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (debvader_amd/) never does.

Conventions: NHWC activations; Conv2D kernels HWIO (kh,kw,cin,cout);
Conv2DTranspose kernels (kh,kw,cout,cin); Dense (in,out).  dtype is a parameter:
float64 for parity checks, float32 when timed as the CPU baseline.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# architecture description (reference: model.py:61-161, train.py:104-107)
# --------------------------------------------------------------------------
@dataclass
class Arch:
    """Layer/parameter plan of create_model_vae (model.py:164-218)."""

    input_shape: Tuple[int, int, int] = (59, 59, 6)
    latent_dim: int = 32
    filters: Sequence[int] = (32, 64, 128, 256)
    kernels: Sequence[int] = (3, 3, 3, 3)
    kl_weight: float = 0.01          # model.py:213
    kl_multiplicity: int = 2         # regulariser applied to both leaves of (distribution, value)
    bn_eps: float = 1e-3             # Keras BatchNormalization default (model.py:79)
    bn_momentum: float = 0.99
    sigma_floor: float = 1e-4        # model.py:156
    diag_shift: float = 1e-5         # model.py:49

    def __post_init__(self):
        H, W, C = self.input_shape
        assert H == W, "square stamps only (reference uses input_shape[0] for both axes, model.py:116,140)"
        self.nb = C
        # params_size of MultivariateNormalTriL (model.py:97): d + d(d+1)/2
        d = self.latent_dim
        self.params_size = d + d * (d + 1) // 2
        # decoder hard-codes params_size(32) for its first Dense (model.py:114)
        self.dec_hidden = 32 + 32 * 33 // 2
        self.w0 = int(np.ceil(H / 2 ** len(self.filters)))      # model.py:116
        # encoder spatial sizes
        self.enc_sizes = [H]
        for _ in self.filters:
            self.enc_sizes.append(-(-self.enc_sizes[-1] // 2))   # SAME stride 2 -> ceil
        self.flat = self.enc_sizes[-1] ** 2 * self.filters[-1]
        self.dec_out = self.w0 * 2 ** len(self.filters)
        crop = self.dec_out - H                                   # model.py:140
        if crop > 0:
            if crop % 2 == 0:
                self.crop = (crop // 2, crop // 2)                 # model.py:143-144
            else:
                self.crop = (crop // 2, crop // 2 + 1)             # model.py:146-148
        else:
            self.crop = (0, 0)

    # ---- parameter table in TF-checkpoint order -------------
    def param_specs(self) -> List[Tuple[str, Tuple[int, ...], bool]]:
        """(name, shape, trainable) for all 64 tensors; encoder first."""
        H, W, C = self.input_shape
        out: List[Tuple[str, Tuple[int, ...], bool]] = []
        out += [("enc/bn/gamma", (C,), True), ("enc/bn/beta", (C,), True),
                ("enc/bn/moving_mean", (C,), False), ("enc/bn/moving_variance", (C,), False)]
        cin = C
        for i, (f, k) in enumerate(zip(self.filters, self.kernels)):
            s_in, s_out = self.enc_sizes[i], self.enc_sizes[i + 1]
            out += [(f"enc/conv{2*i}/kernel", (k, k, cin, f), True), (f"enc/conv{2*i}/bias", (f,), True),
                    (f"enc/prelu{2*i}/alpha", (s_in, s_in, f), True),
                    (f"enc/conv{2*i+1}/kernel", (k, k, f, f), True), (f"enc/conv{2*i+1}/bias", (f,), True),
                    (f"enc/prelu{2*i+1}/alpha", (s_out, s_out, f), True)]
            cin = f
        out += [("enc/prelu_flat/alpha", (self.flat,), True),
                ("enc/dense/kernel", (self.flat, self.params_size), True),
                ("enc/dense/bias", (self.params_size,), True)]
        d = self.latent_dim
        w0, fl = self.w0, self.filters[-1]
        out += [("dec/prelu_in/alpha", (d,), True),
                ("dec/dense0/kernel", (d, self.dec_hidden), True), ("dec/dense0/bias", (self.dec_hidden,), True),
                ("dec/prelu_h/alpha", (self.dec_hidden,), True),
                ("dec/dense1/kernel", (self.dec_hidden, w0 * w0 * fl), True), ("dec/dense1/bias", (w0 * w0 * fl,), True),
                ("dec/prelu_r/alpha", (w0 * w0 * fl,), True)]
        cin = fl
        size = w0
        n = len(self.filters)
        for j, i in enumerate(range(n - 1, -1, -1)):
            f, k = self.filters[i], self.kernels[i]
            size *= 2
            out += [(f"dec/convt{2*j}/kernel", (k, k, f, cin), True), (f"dec/convt{2*j}/bias", (f,), True),
                    (f"dec/prelut{2*j}/alpha", (size, size, f), True),
                    (f"dec/convt{2*j+1}/kernel", (k, k, f, f), True), (f"dec/convt{2*j+1}/bias", (f,), True),
                    (f"dec/prelut{2*j+1}/alpha", (size, size, f), True)]
            cin = f
        out += [("dec/head/kernel", (3, 3, cin, 2 * C), True), ("dec/head/bias", (2 * C,), True)]
        return out

    def param_counts(self) -> Tuple[int, int]:
        enc = sum(int(np.prod(s)) for n, s, _ in self.param_specs() if n.startswith("enc/"))
        dec = sum(int(np.prod(s)) for n, s, _ in self.param_specs() if n.startswith("dec/"))
        return enc, dec


def init_params(arch: Arch, seed: int = 0, perturb: float = 0.0, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Keras default initialisers: Glorot-uniform kernels, zero biases,
    PReLU alpha zeros, BN gamma=1 beta=0 mean=0 var=1.  `perturb` adds N(0,perturb)
    noise to biases/alphas/gamma/beta so that those paths are exercised in tests."""
    rng = np.random.default_rng(seed)
    p: Dict[str, np.ndarray] = {}
    for name, shape, _ in arch.param_specs():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                fan_in, fan_out = rf * shape[2], rf * shape[3]
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, size=shape)
        elif leaf in ("gamma", "moving_variance"):
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        if perturb and leaf in ("bias", "alpha", "gamma", "beta"):
            a = a + rng.normal(0.0, perturb, size=shape)
        if perturb and leaf == "moving_mean":
            a = a + rng.normal(0.0, perturb, size=shape)
        if perturb and leaf == "moving_variance":
            a = a + np.abs(rng.normal(0.0, perturb, size=shape))
        p[name] = a.astype(dtype)
    return p


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def same_pad(n_in: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF SAME: out=ceil(in/s); total=max((out-1)s+k-in,0); before=total//2."""
    n_out = -(-n_in // s)
    tot = max((n_out - 1) * s + k - n_in, 0)
    return n_out, tot // 2, tot - tot // 2


def conv2d_same(x, w, b, stride):
    """Keras Conv2D(padding='same') forward (model.py:81-91,137). x NHWC, w HWIO."""
    N, H, W, Ci = x.shape
    kh_, kw_, _, Co = w.shape
    Ho, pt, pb = same_pad(H, kh_, stride)
    Wo, pl, pr = same_pad(W, kw_, stride)
    xp = np.zeros((N, H + pt + pb, W + pl + pr, Ci), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W] = x
    y = np.zeros((N, Ho, Wo, Co), dtype=x.dtype)
    for kh in range(kh_):
        for kw in range(kw_):
            xs = xp[:, kh:kh + stride * (Ho - 1) + 1:stride, kw:kw + stride * (Wo - 1) + 1:stride]
            y += xs.reshape(-1, Ci).dot(w[kh, kw]).reshape(N, Ho, Wo, Co)
    return y + b


def conv2d_same_bwd(x, w, dy, stride):
    """Gradients of conv2d_same wrt x, w, b."""
    N, H, W, Ci = x.shape
    kh_, kw_, _, Co = w.shape
    Ho, pt, pb = same_pad(H, kh_, stride)
    Wo, pl, pr = same_pad(W, kw_, stride)
    xp = np.zeros((N, H + pt + pb, W + pl + pr, Ci), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    dy2 = dy.reshape(-1, Co)
    for kh in range(kh_):
        for kw in range(kw_):
            sl = (slice(None), slice(kh, kh + stride * (Ho - 1) + 1, stride),
                  slice(kw, kw + stride * (Wo - 1) + 1, stride))
            dw[kh, kw] = xp[sl].reshape(-1, Ci).T.dot(dy2)
            dxp[sl] += dy2.dot(w[kh, kw].T).reshape(N, Ho, Wo, Ci)
    return dxp[:, pt:pt + H, pl:pl + W], dw, dy2.sum(0)


def convt2d_same(x, k, b, stride):
    """Keras Conv2DTranspose(padding='same') forward (model.py:121-134): the
    input-gradient of a SAME conv whose input has size in*stride.
    x (N,Hi,Wi,Ci), k (kh,kw,Co,Ci) -> (N,Hi*s,Wi*s,Co); o = i*s + kh - pad_before."""
    N, Hi, Wi, Ci = x.shape
    kh_, kw_, Co, _ = k.shape
    Ho, Wo = Hi * stride, Wi * stride
    _, pt, pb = same_pad(Ho, kh_, stride)
    _, pl, pr = same_pad(Wo, kw_, stride)
    full = np.zeros((N, Ho + pt + pb, Wo + pl + pr, Co), dtype=x.dtype)
    x2 = x.reshape(-1, Ci)
    for kh in range(kh_):
        for kw in range(kw_):
            full[:, kh:kh + stride * (Hi - 1) + 1:stride, kw:kw + stride * (Wi - 1) + 1:stride] += \
                x2.dot(k[kh, kw].T).reshape(N, Hi, Wi, Co)
    return full[:, pt:pt + Ho, pl:pl + Wo] + b


def convt2d_same_bwd(x, k, dy, stride):
    N, Hi, Wi, Ci = x.shape
    kh_, kw_, Co, _ = k.shape
    Ho, Wo = Hi * stride, Wi * stride
    _, pt, pb = same_pad(Ho, kh_, stride)
    _, pl, pr = same_pad(Wo, kw_, stride)
    dfull = np.zeros((N, Ho + pt + pb, Wo + pl + pr, Co), dtype=x.dtype)
    dfull[:, pt:pt + Ho, pl:pl + Wo] = dy
    dx = np.zeros_like(x)
    dk = np.zeros_like(k)
    x2 = x.reshape(-1, Ci)
    for kh in range(kh_):
        for kw in range(kw_):
            g = dfull[:, kh:kh + stride * (Hi - 1) + 1:stride, kw:kw + stride * (Wi - 1) + 1:stride].reshape(-1, Co)
            dx += g.dot(k[kh, kw]).reshape(N, Hi, Wi, Ci)
            dk[kh, kw] = g.T.dot(x2)
    return dx, dk, dy.reshape(-1, Co).sum(0)


def prelu(u, alpha):
    """Keras PReLU with per-element alpha (model.py:84 etc.)."""
    return np.maximum(u, 0) + alpha * np.minimum(u, 0)


def prelu_bwd(u, alpha, da):
    du = da * np.where(u > 0, 1.0, alpha).astype(u.dtype)
    dalpha = (da * np.minimum(u, 0)).sum(0)
    return du, dalpha


def softplus(x):
    return np.logaddexp(0.0, x)


def sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def fill_triangular(v):
    """tfp.math.fill_triangular (lower), op order pinned by
    deblender_to_onnx.ipynb:160-187: concat(x[n:], reverse(x)) -> reshape(n,n) -> lower band.
    Known answer: [1..6] -> [[4,0,0],[6,5,0],[3,2,1]]."""
    m = v.shape[-1]
    n = int((math.isqrt(1 + 8 * m) - 1) // 2)
    assert n * (n + 1) // 2 == m
    xc = np.concatenate([v[..., n:], v[..., ::-1]], axis=-1)
    y = xc.reshape(v.shape[:-1] + (n, n))
    return np.tril(y)


def fill_triangular_index(n: int) -> np.ndarray:
    """idx[i,j] (j<=i) = position in the length n(n+1)/2 vector that lands at L[i,j]."""
    m = n * (n + 1) // 2
    src = fill_triangular(np.arange(1, m + 1, dtype=np.float64)).astype(np.int64) - 1
    src[np.triu_indices(n, 1)] = -1
    return src


# --------------------------------------------------------------------------
# model forward / backward
# --------------------------------------------------------------------------
def _bn_forward(arch, p, x, training):
    """Keras BatchNormalization over the band axis (model.py:79)."""
    g, b = p["enc/bn/gamma"], p["enc/bn/beta"]
    if training:
        mean = x.mean(axis=(0, 1, 2))
        var = x.var(axis=(0, 1, 2))                  # biased
    else:
        mean, var = p["enc/bn/moving_mean"], p["enc/bn/moving_variance"]
    inv = 1.0 / np.sqrt(var + arch.bn_eps)
    xhat = (x - mean) * inv
    return xhat * g + b, xhat, mean, var


def encoder_forward(arch: Arch, p, x, training=False, cache=None):
    """create_encoder (model.py:61-100) -> (B, params_size)."""
    c = cache if cache is not None else {}
    h, xhat, mean, var = _bn_forward(arch, p, x, training)
    c["xhat"], c["bn_mean"], c["bn_var"] = xhat, mean, var
    c["enc_in0"] = h
    for i in range(len(arch.filters)):
        for j, s in ((2 * i, 1), (2 * i + 1, 2)):
            c[f"enc_in{j}"] = h
            u = conv2d_same(h, p[f"enc/conv{j}/kernel"], p[f"enc/conv{j}/bias"], s)
            c[f"enc_u{j}"] = u
            h = prelu(u, p[f"enc/prelu{j}/alpha"])
    B = x.shape[0]
    hf = h.reshape(B, -1)                                  # Flatten, model.py:94
    c["enc_flat_u"] = hf
    f = prelu(hf, p["enc/prelu_flat/alpha"])               # model.py:95
    c["enc_flat_a"] = f
    t = f.dot(p["enc/dense/kernel"]) + p["enc/dense/bias"]  # model.py:96-98
    return t


def sampler_forward(arch: Arch, t, eps):
    """MultivariateNormalTriL + in-repo restatement MvNormal (model.py:43-58, 211-214):
    mu=t[:d]; L=fill_triangular(t[d:]); diag<-softplus(diag)+1e-5; z=mu+L.eps.
    KL_b (single-sample MC) = log q(z) - log p(z) = 0.5|z|^2 - 0.5|eps|^2 - sum log L_ii."""
    d = arch.latent_dim
    mu = t[:, :d]
    Lraw = fill_triangular(t[:, d:])
    di = np.arange(d)
    L = Lraw.copy()
    L[:, di, di] = softplus(Lraw[:, di, di]) + arch.diag_shift
    z = mu + np.einsum("bij,bj->bi", L, eps)
    kl = 0.5 * (z * z).sum(1) - 0.5 * (eps * eps).sum(1) - np.log(L[:, di, di]).sum(1)
    return mu, L, Lraw, z, kl


def decoder_forward(arch: Arch, p, z, cache=None):
    """create_decoder (model.py:103-161) up to the Cropping2D; returns (loc, scale)."""
    c = cache if cache is not None else {}
    B = z.shape[0]
    c["dec_z"] = z
    h = prelu(z, p["dec/prelu_in/alpha"])                          # model.py:113
    c["dec_a_in"] = h
    u = h.dot(p["dec/dense0/kernel"]) + p["dec/dense0/bias"]       # model.py:114
    c["dec_u_h"] = u
    h = prelu(u, p["dec/prelu_h/alpha"])                           # model.py:115
    c["dec_a_h"] = h
    u = h.dot(p["dec/dense1/kernel"]) + p["dec/dense1/bias"]       # model.py:117
    c["dec_u_r"] = u
    h = prelu(u, p["dec/prelu_r/alpha"])                           # model.py:118
    h = h.reshape(B, arch.w0, arch.w0, arch.filters[-1])           # model.py:119
    for j in range(2 * len(arch.filters)):
        s = 2 if j % 2 == 0 else 1                                 # model.py:121-134
        c[f"dec_in{j}"] = h
        u = convt2d_same(h, p[f"dec/convt{j}/kernel"], p[f"dec/convt{j}/bias"], s)
        c[f"dec_u{j}"] = u
        h = prelu(u, p[f"dec/prelut{j}/alpha"])
    c["head_in"] = h
    tpre = conv2d_same(h, p["dec/head/kernel"], p["dec/head/bias"], 1)   # model.py:137
    c["head_pre"] = tpre
    t = np.maximum(tpre, 0)                                        # activation="relu"
    c0, c1 = arch.crop
    H = arch.input_shape[0]
    t = t[:, c0:c0 + H, c0:c0 + H, :]                              # model.py:140-148
    nb = arch.nb
    loc = t[..., :nb]
    scale = arch.sigma_floor + t[..., nb:]                         # model.py:154-157
    return loc, scale


def normal_nll(y, loc, scale):
    """vae_loss (metrics.py:16-26) = -Normal(loc,scale).log_prob(y), per pixel and band."""
    zs = (y - loc) / scale
    return 0.5 * zs * zs + np.log(scale) + 0.5 * LOG_2PI


def forward(arch: Arch, p, x, eps, training=False):
    """net(x) (model.py:216): returns dict with t, mu, L, z, kl, loc, scale and the cache."""
    c: Dict[str, np.ndarray] = {}
    t = encoder_forward(arch, p, x, training, c)
    mu, L, Lraw, z, kl = sampler_forward(arch, t, eps)
    loc, scale = decoder_forward(arch, p, z, c)
    c.update(t=t, mu=mu, L=L, Lraw=Lraw, z=z, kl=kl, loc=loc, scale=scale, eps=eps, x=x)
    return c


def losses(arch: Arch, c, y, global_batch: Optional[int] = None):
    """Keras total loss for compile(loss=vae_loss) + activity regulariser (train.py:125-130):
    nll_mean = mean over all B*H*W*C elements (SUM_OVER_BATCH_SIZE on the 4-D tensor);
    kl_reg   = k * weight * mean_b(KL_b) / B   (SURVEY A7).
    With `global_batch` the partial (per-shard) sums use the global normalisers."""
    B = y.shape[0]
    Bg = global_batch or B
    nll = normal_nll(y, c["loc"], c["scale"])
    npix = int(np.prod(y.shape[1:]))
    nll_mean = nll.sum() / (Bg * npix)
    kl_reg = arch.kl_multiplicity * arch.kl_weight * c["kl"].sum() / (Bg * Bg)
    mse = ((y - c["loc"]) ** 2).sum() / (Bg * npix)
    return dict(loss=nll_mean + kl_reg, nll_mean=nll_mean, kl_reg=kl_reg, mse=mse, nll=nll)


def backward(arch: Arch, p, c, y, global_batch: Optional[int] = None, train_decoder=True):
    """Analytic gradient of losses()['loss'] wrt every trainable tensor."""
    g: Dict[str, np.ndarray] = {}
    B = y.shape[0]
    Bg = global_batch or B
    H, nb = arch.input_shape[0], arch.nb
    npix = int(np.prod(y.shape[1:]))
    loc, scale = c["loc"], c["scale"]
    inv = 1.0 / scale
    r = (y - loc) * inv
    dloc = -(r * inv) / (Bg * npix)
    dscale = (inv - r * r * inv) / (Bg * npix)
    dt = np.zeros_like(c["head_pre"])
    c0 = arch.crop[0]
    dt[:, c0:c0 + H, c0:c0 + H, :nb] = dloc
    dt[:, c0:c0 + H, c0:c0 + H, nb:] = dscale
    dt = dt * (c["head_pre"] > 0)
    dh, dw, db = conv2d_same_bwd(c["head_in"], p["dec/head/kernel"], dt, 1)
    g["dec/head/kernel"], g["dec/head/bias"] = dw, db
    for j in range(2 * len(arch.filters) - 1, -1, -1):
        s = 2 if j % 2 == 0 else 1
        du, dal = prelu_bwd(c[f"dec_u{j}"], p[f"dec/prelut{j}/alpha"], dh)
        g[f"dec/prelut{j}/alpha"] = dal
        dh, dk, db = convt2d_same_bwd(c[f"dec_in{j}"], p[f"dec/convt{j}/kernel"], du, s)
        g[f"dec/convt{j}/kernel"], g[f"dec/convt{j}/bias"] = dk, db
    dh = dh.reshape(B, -1)
    du, dal = prelu_bwd(c["dec_u_r"], p["dec/prelu_r/alpha"], dh)
    g["dec/prelu_r/alpha"] = dal
    g["dec/dense1/kernel"] = c["dec_a_h"].T.dot(du)
    g["dec/dense1/bias"] = du.sum(0)
    dh = du.dot(p["dec/dense1/kernel"].T)
    du, dal = prelu_bwd(c["dec_u_h"], p["dec/prelu_h/alpha"], dh)
    g["dec/prelu_h/alpha"] = dal
    g["dec/dense0/kernel"] = c["dec_a_in"].T.dot(du)
    g["dec/dense0/bias"] = du.sum(0)
    dh = du.dot(p["dec/dense0/kernel"].T)
    dz, dal = prelu_bwd(c["dec_z"], p["dec/prelu_in/alpha"], dh)
    g["dec/prelu_in/alpha"] = dal
    if not train_decoder:
        g = {}
    # sampler + KL
    d = arch.latent_dim
    kls = arch.kl_multiplicity * arch.kl_weight / (Bg * Bg)
    z, eps, L, Lraw = c["z"], c["eps"], c["L"], c["Lraw"]
    dz = dz + kls * z
    dmu = dz
    dL = np.einsum("bi,bj->bij", dz, eps)
    di = np.arange(d)
    dL[:, di, di] -= kls / L[:, di, di]
    dL[:, di, di] *= sigmoid(Lraw[:, di, di])
    idx = fill_triangular_index(d)
    dt_ = np.zeros_like(c["t"])
    dt_[:, :d] = dmu
    ii, jj = np.tril_indices(d)
    dt_[:, d + idx[ii, jj]] = dL[:, ii, jj]
    g["enc/dense/kernel"] = c["enc_flat_a"].T.dot(dt_)
    g["enc/dense/bias"] = dt_.sum(0)
    dh = dt_.dot(p["enc/dense/kernel"].T)
    dh, dal = prelu_bwd(c["enc_flat_u"], p["enc/prelu_flat/alpha"], dh)
    g["enc/prelu_flat/alpha"] = dal
    n = len(arch.filters)
    s_last = arch.enc_sizes[-1]
    dh = dh.reshape(B, s_last, s_last, arch.filters[-1])
    for j in range(2 * n - 1, -1, -1):
        s = 2 if j % 2 == 1 else 1
        du, dal = prelu_bwd(c[f"enc_u{j}"], p[f"enc/prelu{j}/alpha"], dh)
        g[f"enc/prelu{j}/alpha"] = dal
        dh, dw, db = conv2d_same_bwd(c[f"enc_in{j}"], p[f"enc/conv{j}/kernel"], du, s)
        g[f"enc/conv{j}/kernel"], g[f"enc/conv{j}/bias"] = dw, db
    g["enc/bn/gamma"] = (dh * c["xhat"]).sum(axis=(0, 1, 2))
    g["enc/bn/beta"] = dh.sum(axis=(0, 1, 2))
    return g


def bn_moving_update(arch: Arch, p, c, n_global: Optional[int] = None, unbiased=True):
    """Keras moving-statistics update in training: moving = moving*m + batch*(1-m);
    the fused path feeds the Bessel-corrected variance."""
    m = arch.bn_momentum
    n = n_global or int(np.prod(c["x"].shape[:3]))
    var = c["bn_var"] * (n / (n - 1.0)) if unbiased else c["bn_var"]
    p["enc/bn/moving_mean"] = p["enc/bn/moving_mean"] * m + c["bn_mean"] * (1 - m)
    p["enc/bn/moving_variance"] = p["enc/bn/moving_variance"] * m + var * (1 - m)


@dataclass
class AdamState:
    """tf.optimizers.legacy.Adam (train.py:126)."""
    lr: float = 1e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-7
    t: int = 0
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)


def adam_step(st: AdamState, p, g):
    st.t += 1
    lr_t = st.lr * math.sqrt(1.0 - st.b2 ** st.t) / (1.0 - st.b1 ** st.t)
    for k, gk in g.items():
        if k not in st.m:
            st.m[k] = np.zeros_like(p[k])
            st.v[k] = np.zeros_like(p[k])
        st.m[k] += (gk - st.m[k]) * (1 - st.b1)
        st.v[k] += (gk * gk - st.v[k]) * (1 - st.b2)
        p[k] = p[k] - lr_t * st.m[k] / (np.sqrt(st.v[k]) + st.eps)


def train_step(arch: Arch, p, st: AdamState, x, y, eps, train_decoder=True):
    """One Keras train_function call (SURVEY 3.1 hot loop): fwd(training) + loss + bwd + Adam + BN moving update."""
    c = forward(arch, p, x, eps, training=True)
    out = losses(arch, c, y)
    g = backward(arch, p, c, y, train_decoder=train_decoder)
    adam_step(st, p, g)
    bn_moving_update(arch, p, c)
    out.pop("nll")
    return out, g


# --------------------------------------------------------------------------
# Philox reference for the engine's own eps generator
# --------------------------------------------------------------------------
def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox4x32-10 (Salmon et al. 2011). counter (...,4) uint32, key (...,2) uint32."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = [counter[..., i].astype(np.uint32) for i in range(4)]
    k0 = key[..., 0].astype(np.uint32).copy()
    k1 = key[..., 1].astype(np.uint32).copy()
    for _ in range(10):
        p0 = M0 * c[0].astype(np.uint64)
        p1 = M1 * c[2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + W0).astype(np.uint32)
        k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_normal(seed: int, stream: int, n_rows: int, n_cols: int) -> np.ndarray:
    """The engine's own eps generator (TF's Philox stream is not reproducible, so the
    engine defines its own): element (row, col) uses counter (row, col//4, stream, 0), key (seed lo, seed hi);
    u32 -> uniform (0,1] -> Box-Muller pairs (0,1),(2,3) -> float32."""
    rows = np.arange(n_rows, dtype=np.uint32)[:, None]
    blocks = np.arange((n_cols + 3) // 4, dtype=np.uint32)[None, :]
    ctr = np.zeros((n_rows, blocks.shape[1], 4), dtype=np.uint32)
    ctr[..., 0] = rows
    ctr[..., 1] = blocks
    ctr[..., 2] = np.uint32(stream)
    key = np.zeros((n_rows, blocks.shape[1], 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, key)
    u = (r.astype(np.float64) + 1.0) * (1.0 / 4294967296.0)      # (0,1]
    out = np.empty((n_rows, blocks.shape[1], 4), dtype=np.float64)
    for a in (0, 2):
        rad = np.sqrt(-2.0 * np.log(u[..., a]))
        ang = 2.0 * np.pi * u[..., a + 1]
        out[..., a] = rad * np.cos(ang)
        out[..., a + 1] = rad * np.sin(ang)
    return out.reshape(n_rows, -1)[:, :n_cols].astype(np.float32)
