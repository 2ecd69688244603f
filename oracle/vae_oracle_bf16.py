"""bf16-storage variant of the CPU oracle (TEST INFRASTRUCTURE, not product; PARITY UNPINNED like vae_oracle.py).

The reference (astrodeepnet/debvader) computes in float32 only.  BASELINE configs[2] asks for the same model with
bf16 storage / bf16 MFMA operands; the engine's bf16 kernel family (debvader_amd/csrc/bf16.h) rounds to bfloat16 at
fixed points of the pipeline.  This module restates vae_oracle.forward / backward with a round-to-nearest-even bf16
rounding at exactly those points (and float64 arithmetic everywhere else), so that tests can tell a kernel bug (tight
tolerance against THIS file) from the precision loss of the format (loose, stated tolerance against vae_oracle, the
restatement of model.py:61-161 / metrics.py:16-26).

Rounding points (engine_bf16.inl): normalised input x-hat; every conv / conv-transpose weight matrix (first conv with
the input BatchNorm folded in: W*gamma, and sum_c W*beta on a constant-one channel); pre-activation u and PReLU output
a of every conv / conv-transpose layer; the decoder trunk's output where it enters the conv-transpose stack; the
gradient w.r.t. the head's pre-activation; every d(pre-activation) of the conv stacks; the two data gradients that
leave / enter the bf16 stacks (decoder input, encoder output).  fp32 in the engine, float64 here: dense trunk,
sampler, head arithmetic, all accumulations.

Round 6 (debvader_amd/csrc/btrunk.hip): when the last encoder level has a multiple of 64 filters (`trunk_on_mfma`: the
59-px and 128-px nets) the two large Dense layers of the trunk and their gradients are bf16-MFMA products as well.
Additional rounding points then: the two Dense kernels (enc/dense, dec/dense1) in all three products they enter; the
flatten PReLU's output, the hidden layer's PReLU output and d(t) where they enter a product; the pre-activation of the
trunk's output where it is stored for its PReLU backward; d(pre-activation) of the trunk's output and of the last
encoder conv (both now leave a fused epilogue: no separate rounding of d(activation) in between).

Only tests/ may import this module.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import vae_oracle as vo


def bf16(x):
    """Round to nearest-even bfloat16 (what v_cvt_pk_bf16_f32 does), returned in x's dtype."""
    x = np.asarray(x)
    a = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((a.astype(np.uint64) + 0x7FFF + ((a >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(x.dtype if x.dtype.kind == "f" else np.float64)


def _folded_first_conv(arch, p, xhat_r):
    """First conv with the input BatchNorm folded into a (C+1)-channel kernel: channels 0..C-1 = W*gamma on x-hat,
    channel C = sum_c W*beta on a constant one that exists inside the image only (SAME zero padding after BN)."""
    W, g, b = p["enc/conv0/kernel"], p["enc/bn/gamma"], p["enc/bn/beta"]
    w = np.concatenate([bf16(W * g[None, None, :, None]), bf16((W * b[None, None, :, None]).sum(2, keepdims=True))], axis=2)
    xin = np.concatenate([xhat_r, np.ones(xhat_r.shape[:3] + (1,), xhat_r.dtype)], axis=3)
    return xin, w


def trunk_on_mfma(arch: vo.Arch) -> bool:
    """engine_bf16.inl::bf_alloc: the dense trunk runs on the bf16 matrix cores when the stamp-inner tensor it reads has
    a multiple of 64 channels"""
    return arch.filters[-1] % 64 == 0


def forward(arch: vo.Arch, p, x, eps, training=False, trunk=None):
    c: Dict[str, np.ndarray] = {}
    trunk = trunk_on_mfma(arch) if trunk is None else trunk
    c["_trunk"] = trunk
    _, xhat, mean, var = vo._bn_forward(arch, p, x, training)
    c["xhat"], c["bn_mean"], c["bn_var"] = xhat, mean, var
    xin, w0 = _folded_first_conv(arch, p, bf16(xhat))
    c["enc_in0"], c["enc_w0"] = xin, w0
    h = xin
    for j in range(2 * len(arch.filters)):
        s = 2 if j % 2 else 1
        w = w0 if j == 0 else bf16(p[f"enc/conv{j}/kernel"])
        c[f"enc_in{j}"] = h
        u32 = vo.conv2d_same(h, w, p[f"enc/conv{j}/bias"], s)
        c[f"enc_u{j}"] = bf16(u32)
        h = bf16(vo.prelu(u32, p[f"enc/prelu{j}/alpha"]))
    B = x.shape[0]
    hf = h.reshape(B, -1)
    c["enc_flat_u"] = hf
    f = vo.prelu(hf, p["enc/prelu_flat/alpha"])
    if trunk:
        f = bf16(f)                                   # A fragments of the product: PReLU on load, rounded again
    c["enc_flat_a"] = f
    t = f.dot(bf16(p["enc/dense/kernel"]) if trunk else p["enc/dense/kernel"]) + p["enc/dense/bias"]
    mu, L, Lraw, z, kl = vo.sampler_forward(arch, t, eps)
    # decoder trunk (fp32 in the engine)
    c["dec_z"] = z
    hd = vo.prelu(z, p["dec/prelu_in/alpha"])
    c["dec_a_in"] = hd
    u = hd.dot(p["dec/dense0/kernel"]) + p["dec/dense0/bias"]
    c["dec_u_h"] = u
    hd = vo.prelu(u, p["dec/prelu_h/alpha"])
    if trunk:
        hd = bf16(hd)                                 # fp32 rows rounded where the product loads them
    c["dec_a_h"] = hd
    u = hd.dot(bf16(p["dec/dense1/kernel"]) if trunk else p["dec/dense1/kernel"]) + p["dec/dense1/bias"]
    c["dec_u_r"] = bf16(u) if trunk else u            # stored in bf16 (stamp-inner) for the PReLU backward
    hd = bf16(vo.prelu(u, p["dec/prelu_r/alpha"])).reshape(B, arch.w0, arch.w0, arch.filters[-1])
    for j in range(2 * len(arch.filters)):
        s = 2 if j % 2 == 0 else 1
        c[f"dec_in{j}"] = hd
        u32 = vo.convt2d_same(hd, bf16(p[f"dec/convt{j}/kernel"]), p[f"dec/convt{j}/bias"], s)
        c[f"dec_u{j}"] = bf16(u32)
        hd = bf16(vo.prelu(u32, p[f"dec/prelut{j}/alpha"]))
    c["head_in"] = hd
    tpre = vo.conv2d_same(hd, bf16(p["dec/head/kernel"]), p["dec/head/bias"], 1)
    c["head_pre"] = tpre
    tt = np.maximum(tpre, 0)
    c0 = arch.crop[0]
    H, nb = arch.input_shape[0], arch.nb
    tt = tt[:, c0:c0 + H, c0:c0 + H, :]
    loc, scale = tt[..., :nb], arch.sigma_floor + tt[..., nb:]
    c.update(t=t, mu=mu, L=L, Lraw=Lraw, z=z, kl=kl, loc=loc, scale=scale, eps=eps, x=x)
    return c


def _prelu_bwd(u_r, alpha, dA32, fused):
    """fused: the data-gradient epilogue holds dA in fp32; unfused: dA went through a bf16 store first."""
    dA = dA32 if fused else bf16(dA32)
    du32 = dA * np.where(u_r > 0, 1.0, alpha)
    dalpha = (dA * np.minimum(u_r, 0)).sum(0)
    db = du32.sum(axis=(0, 1, 2))
    return bf16(du32), dalpha, db


def backward(arch: vo.Arch, p, c, y, global_batch: Optional[int] = None, train_decoder=True, fused=True):
    """Gradient of vae_oracle.losses()['loss'] as the bf16 engine computes it.  `fused`: the stamp count is padded to a
    multiple of 64, so the PReLU backward runs inside the data-gradient epilogues."""
    g: Dict[str, np.ndarray] = {}
    B = y.shape[0]
    Bg = global_batch or B
    H, nb = arch.input_shape[0], arch.nb
    npix = int(np.prod(y.shape[1:]))
    loc, scale = c["loc"], c["scale"]
    inv = 1.0 / scale
    r = (y - loc) * inv
    dt = np.zeros_like(c["head_pre"])
    c0 = arch.crop[0]
    dt[:, c0:c0 + H, c0:c0 + H, :nb] = -(r * inv) / (Bg * npix)
    dt[:, c0:c0 + H, c0:c0 + H, nb:] = (inv - r * r * inv) / (Bg * npix)
    dt = bf16(dt * (c["head_pre"] > 0))
    dh, dw, db = vo.conv2d_same_bwd(c["head_in"], bf16(p["dec/head/kernel"]), dt, 1)
    g["dec/head/kernel"], g["dec/head/bias"] = dw, db
    n2 = 2 * len(arch.filters)
    for j in range(n2 - 1, -1, -1):
        s = 2 if j % 2 == 0 else 1
        du, dal, db = _prelu_bwd(c[f"dec_u{j}"], p[f"dec/prelut{j}/alpha"], dh, fused)
        g[f"dec/prelut{j}/alpha"], g[f"dec/convt{j}/bias"] = dal, db
        dh, dk, _ = vo.convt2d_same_bwd(c[f"dec_in{j}"], bf16(p[f"dec/convt{j}/kernel"]), du, s)
        g[f"dec/convt{j}/kernel"] = dk
    trunk = c.get("_trunk", False)
    if trunk:
        # the PReLU backward of the trunk's output runs in the epilogue of the first transposed conv's data gradient
        dA = (dh if fused else bf16(dh)).reshape(B, -1)
        ur = c["dec_u_r"]
        du32 = dA * np.where(ur > 0, 1.0, p["dec/prelu_r/alpha"])
        g["dec/prelu_r/alpha"] = (dA * np.minimum(ur, 0)).sum(0)
        g["dec/dense1/bias"] = du32.sum(0)
        du = bf16(du32)
        g["dec/dense1/kernel"] = c["dec_a_h"].T.dot(du)
        dh = du.dot(bf16(p["dec/dense1/kernel"]).T)
    else:
        dh = bf16(dh).reshape(B, -1)
        du, dal = vo.prelu_bwd(c["dec_u_r"], p["dec/prelu_r/alpha"], dh)
        g["dec/prelu_r/alpha"] = dal
        g["dec/dense1/kernel"] = c["dec_a_h"].T.dot(du)
        g["dec/dense1/bias"] = du.sum(0)
        dh = du.dot(p["dec/dense1/kernel"].T)
    du, dal = vo.prelu_bwd(c["dec_u_h"], p["dec/prelu_h/alpha"], dh)
    g["dec/prelu_h/alpha"] = dal
    g["dec/dense0/kernel"] = c["dec_a_in"].T.dot(du)
    g["dec/dense0/bias"] = du.sum(0)
    dh = du.dot(p["dec/dense0/kernel"].T)
    dz, dal = vo.prelu_bwd(c["dec_z"], p["dec/prelu_in/alpha"], dh)
    g["dec/prelu_in/alpha"] = dal
    if not train_decoder:
        g = {}
    d = arch.latent_dim
    kls = arch.kl_multiplicity * arch.kl_weight / (Bg * Bg)
    z, eps, L, Lraw = c["z"], c["eps"], c["L"], c["Lraw"]
    dz = dz + kls * z
    dL = np.einsum("bi,bj->bij", dz, eps)
    di = np.arange(d)
    dL[:, di, di] -= kls / L[:, di, di]
    dL[:, di, di] *= vo.sigmoid(Lraw[:, di, di])
    idx = vo.fill_triangular_index(d)
    dt_ = np.zeros_like(c["t"])
    dt_[:, :d] = dz
    ii, jj = np.tril_indices(d)
    dt_[:, d + idx[ii, jj]] = dL[:, ii, jj]
    g["enc/dense/bias"] = dt_.sum(0)
    if trunk:
        g["enc/dense/kernel"] = c["enc_flat_a"].T.dot(bf16(dt_))
        dh = bf16(dt_).dot(bf16(p["enc/dense/kernel"]).T)
    else:
        g["enc/dense/kernel"] = c["enc_flat_a"].T.dot(dt_)
        dh = dt_.dot(p["enc/dense/kernel"].T)
    dh, dal = vo.prelu_bwd(c["enc_flat_u"], p["enc/prelu_flat/alpha"], dh)
    g["enc/prelu_flat/alpha"] = dal
    s_last = arch.enc_sizes[-1]
    dh = dh.reshape(B, s_last, s_last, arch.filters[-1])
    for j in range(n2 - 1, -1, -1):
        s = 2 if j % 2 == 1 else 1
        # the last encoder layer's PReLU backward is always the separate pass (its d(activation) arrives as fp32 rows)
        # (... unless the trunk runs on the matrix cores: then both PReLU gates of the seam sit in one fp32 epilogue)
        du, dal, db = _prelu_bwd(c[f"enc_u{j}"], p[f"enc/prelu{j}/alpha"], dh, (fused and j != n2 - 1) or (trunk and j == n2 - 1))
        g[f"enc/prelu{j}/alpha"], g[f"enc/conv{j}/bias"] = dal, db
        w = c["enc_w0"] if j == 0 else bf16(p[f"enc/conv{j}/kernel"])
        dh, dw, _ = vo.conv2d_same_bwd(c[f"enc_in{j}"], w, du, s)
        if j > 0:
            g[f"enc/conv{j}/kernel"] = dw
    # unfold the first conv: d(kernel), d(gamma), d(beta) from the gradient of the folded (C+1)-channel kernel
    W, gam, bet = p["enc/conv0/kernel"], p["enc/bn/gamma"], p["enc/bn/beta"]
    C = arch.nb
    g["enc/conv0/kernel"] = dw[:, :, :C, :] * gam[None, None, :, None] + dw[:, :, C:C + 1, :] * bet[None, None, :, None]
    g["enc/bn/gamma"] = (dw[:, :, :C, :] * W).sum(axis=(0, 1, 3))
    g["enc/bn/beta"] = (dw[:, :, C:C + 1, :] * W).sum(axis=(0, 1, 3))
    return g
