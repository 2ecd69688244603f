"""Per-layer parity of the fp32 engine at the REAL layer shapes and the quoted batch (VERDICT r3 "what's weak" #2 (iii)):
every Conv2D / Conv2DTranspose layer of the 59 x 59 x 6 net (model.py:79-98,112-137) at 256 stamps, and of the
128 x 128 x 6 / six-level net at its per-GPU batch of 64 (BASELINE configs[3]), re-computed ALONE by the oracle's float64 primitives from the ENGINE's stored input
of that layer (teacher forcing: nothing cascades, a failing layer is named), forward and backward:

  forward   u = conv(engine's activation below) + bias, a = PReLU(u)          (Winograd / strip / stride-2 / gather-GEMM kernels)
  backward  d(pre-activation) = d(activation) * gate(u)                        (prelu_bwd_kernel; fused in the first layer)
            d(alpha), d(bias), d(kernel) from the engine's d(pre-activation)   (Winograd / strip / tiled weight-gradient kernels)
            d(activation below) = data gradient of the engine's d(pre-activation)

Stated tolerances (fp32 sums of up to 2304 products against float64): activations and data gradients <= 2e-5 * max of
the tensor, kernel gradients (sums over 256 x H x W pixels) <= 1e-4 * max, d(alpha) <= 1e-4 * max, d(bias) <= 2e-4 * max
(a sum over every pixel of every stamp with cancellation).  Measured values are printed with -s.
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu

TOL_ACT, TOL_W, TOL_ALPHA, TOL_BIAS = 2e-5, 1e-4, 1e-4, 2e-4


def _relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _gate(u, alpha):
    return np.where(u > 0, 1.0, alpha)


def _check(report, what, got, ref, tol):
    err = _relmax(got, ref)
    report.append((what, err))
    assert err <= tol, (what, err, tol)


ARCHS = {
    "59px": dict(),                                                             # train.py:104-107
    "128px": dict(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6),
    "59px-5x5": dict(kernels=(5, 5, 5, 5)),
}


@pytest.mark.parametrize("arch_name,B", [("59px", 256), ("128px", 64), ("59px-5x5", 24)])
def test_every_conv_layer_alone_against_the_oracle_primitives_fp32(arch_name, B):
    from debvader_amd import engine as E
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch(**ARCHS[arch_name])
    L2 = 2 * len(arch.filters)
    p = vo.init_params(arch, seed=3, perturb=0.05)
    p["dec/head/bias"][arch.nb:] += 0.3
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    x, y = synthetic_stamps(B, seed=11, size=arch.input_shape[0], nb=arch.nb)
    eps = np.random.default_rng(5).normal(size=(B, arch.latent_dim)).astype(np.float32)
    eng = E.Engine(E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels), max_batch=B))
    eng.set_params(p)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    eng.grad_step(0, first=0, B=B, eps=eps)
    H, C = arch.input_shape[0], arch.nb
    report = []

    def act(name, shape):
        return eng.activation(name, shape).astype(np.float64)

    # first layer, 3x3, rows the strip kernel takes (59 px): its PReLU backward is fused into the weight-gradient kernel
    # (wgrad_strip8_kernel<true>), d(pre-activation) is never materialised and what the pass leaves is d(activation)
    from debvader_amd._lib import DvError
    try:
        eng.activation("enc_du0", (B, H, H, arch.filters[0]))
        first_fused = False
    except DvError:
        first_fused = True

    def enc_geom(j):
        lvl = j // 2
        hout = arch.enc_sizes[lvl + 1] if j % 2 else arch.enc_sizes[lvl]
        return hout, arch.filters[lvl], (2 if j % 2 else 1)

    def dec_geom(j):
        lvl = len(arch.filters) - 1 - j // 2
        hout = arch.w0 * 2 ** (j // 2 + 1)
        return hout, arch.filters[lvl], (2 if j % 2 == 0 else 1)

    # ---------------- encoder, forward ----------------
    # the engine's first conv reads [x-hat (C bands), 1, 0 ...] (8 channels) through a kernel with the BatchNorm folded in:
    # W * gamma on x-hat, sum_c W * beta on the constant one that exists inside the image only
    xn = act("xn", (B, H, H, 8))
    assert np.abs(xn[..., C + 1:]).max() == 0.0 and np.all(xn[..., C] == 1.0)
    W0, gam, bet = p["enc/conv0/kernel"], p["enc/bn/gamma"], p["enc/bn/beta"]
    w0 = np.concatenate([W0 * gam[None, None, :, None], (W0 * bet[None, None, :, None]).sum(2, keepdims=True)], axis=2)
    w0 = w0.astype(np.float32).astype(np.float64)              # the folded kernel is stored in fp32
    enc_in = [None] * L2
    h = xn[..., :C + 1]
    for j in range(L2):
        hout, cout, s = enc_geom(j)
        w = w0 if j == 0 else p[f"enc/conv{j}/kernel"]
        enc_in[j] = h
        u64 = vo.conv2d_same(h, w, p[f"enc/conv{j}/bias"], s)
        u = act(f"enc_u{j}", (B, hout, hout, cout))
        a = act(f"enc_a{j}", (B, hout, hout, cout))
        _check(report, f"enc_u{j}", u, u64, TOL_ACT)
        _check(report, f"enc_a{j}", a, vo.prelu(u, p[f"enc/prelu{j}/alpha"]), 1e-6)      # from the ENGINE's u: one multiply
        h = a                                                  # teacher forcing
        del u64
    # ---------------- decoder, forward ----------------
    fl = arch.filters[-1]
    h = act("dec_in", (B, arch.w0, arch.w0, fl))
    dec_in = [None] * L2
    for j in range(L2):
        hout, cout, s = dec_geom(j)
        dec_in[j] = h
        u64 = vo.convt2d_same(h, p[f"dec/convt{j}/kernel"], p[f"dec/convt{j}/bias"], s)
        u = act(f"dec_u{j}", (B, hout, hout, cout))
        a = act(f"dec_a{j}", (B, hout, hout, cout))
        _check(report, f"dec_u{j}", u, u64, TOL_ACT)
        _check(report, f"dec_a{j}", a, vo.prelu(u, p[f"dec/prelut{j}/alpha"]), 1e-6)
        h = a
        del u64
    head_in = h
    tpre = vo.conv2d_same(head_in, p["dec/head/kernel"], p["dec/head/bias"], 1)
    _check(report, "head_pre", act("head_pre", (B, arch.dec_out, arch.dec_out, 2 * C)), tpre, TOL_ACT)
    del tpre

    # ---------------- backward: head, decoder ----------------
    C2p = (2 * C + 15) // 16 * 16
    dt = act("d_head_pre", (B, arch.dec_out, arch.dec_out, C2p))
    assert np.abs(dt[..., 2 * C:]).max() == 0.0
    dt = dt[..., :2 * C]
    dh, dw, db = vo.conv2d_same_bwd(head_in, p["dec/head/kernel"], dt, 1)
    _check(report, "dec/head/kernel", eng.get_grad("dec/head/kernel"), dw, TOL_W)
    _check(report, "dec/head/bias", eng.get_grad("dec/head/bias"), db, TOL_BIAS)
    for j in range(L2 - 1, -1, -1):
        hout, cout, s = dec_geom(j)
        u = act(f"dec_u{j}", (B, hout, hout, cout))
        alpha = p[f"dec/prelut{j}/alpha"]
        du = act(f"dec_du{j}", (B, hout, hout, cout))
        _check(report, f"dec_du{j}", du, dh * _gate(u, alpha), TOL_ACT)
        _check(report, f"dec/prelut{j}/alpha", eng.get_grad(f"dec/prelut{j}/alpha"), (dh * np.minimum(u, 0)).sum(0), TOL_ALPHA)
        _check(report, f"dec/convt{j}/bias", eng.get_grad(f"dec/convt{j}/bias"), du.sum((0, 1, 2)), TOL_BIAS)
        dh, dk, _ = vo.convt2d_same_bwd(dec_in[j], p[f"dec/convt{j}/kernel"], du, s)       # from the ENGINE's du
        _check(report, f"dec/convt{j}/kernel", eng.get_grad(f"dec/convt{j}/kernel"), dk, TOL_W)
        dec_in[j] = None
        del u, du

    # ---------------- backward: encoder (from the engine's d(pre-activation) of the last conv downwards) ----------------
    for j in range(L2 - 1, 0, -1):
        hout, cout, s = enc_geom(j)
        du = act(f"enc_du{j}", (B, hout, hout, cout))
        dh, dw, _ = vo.conv2d_same_bwd(enc_in[j], p[f"enc/conv{j}/kernel"], du, s)
        _check(report, f"enc/conv{j}/kernel", eng.get_grad(f"enc/conv{j}/kernel"), dw, TOL_W)
        _check(report, f"enc/conv{j}/bias", eng.get_grad(f"enc/conv{j}/bias"), du.sum((0, 1, 2)), TOL_BIAS)
        hp, cp, _ = enc_geom(j - 1)
        u = act(f"enc_u{j - 1}", (B, hp, hp, cp))
        alpha = p[f"enc/prelu{j - 1}/alpha"]
        if j - 1 > 0 or not first_fused:
            _check(report, f"enc_du{j - 1}", act(f"enc_du{j - 1}", (B, hp, hp, cp)), dh * _gate(u, alpha), TOL_ACT)
        else:
            _check(report, "enc_da0", act("enc_da0", (B, hp, hp, cp)), dh, TOL_ACT)
        _check(report, f"enc/prelu{j - 1}/alpha", eng.get_grad(f"enc/prelu{j - 1}/alpha"), (dh * np.minimum(u, 0)).sum(0), TOL_ALPHA)
        enc_in[j] = None
        del u, du
    # first conv + input BatchNorm: d(folded kernel) -> d(kernel), d(gamma), d(beta)
    hp, cp, _ = enc_geom(0)
    if not first_fused:
        du0 = act("enc_du0", (B, hp, hp, cp))
    else:
        du0 = act("enc_da0", (B, hp, hp, cp)) * _gate(act("enc_u0", (B, hp, hp, cp)), p["enc/prelu0/alpha"])
    _, dwf, db0 = vo.conv2d_same_bwd(enc_in[0], w0, du0, 1)
    _check(report, "enc/conv0/bias", eng.get_grad("enc/conv0/bias"), db0, TOL_BIAS)
    gk = dwf[:, :, :C, :] * gam[None, None, :, None] + dwf[:, :, C:C + 1, :] * bet[None, None, :, None]
    _check(report, "enc/conv0/kernel", eng.get_grad("enc/conv0/kernel"), gk, TOL_W)
    _check(report, "enc/bn/gamma", eng.get_grad("enc/bn/gamma"), (dwf[:, :, :C, :] * W0).sum((0, 1, 3)), 2e-3)
    _check(report, "enc/bn/beta", eng.get_grad("enc/bn/beta"), (dwf[:, :, C:C + 1, :] * W0).sum((0, 1, 3)), 2e-3)
    eng.close()
    from tests import margins
    tol_of = lambda n: (TOL_ACT if ("_u" in n or "_du" in n or "_da" in n or n == "head_pre") else 1e-6 if "_a" in n else
                        2e-3 if "/bn/" in n else TOL_BIAS if n.endswith("bias") else TOL_ALPHA if n.endswith("alpha") else TOL_W)
    margins.record(f"fp32 engine, every conv layer ALONE (teacher-forced) vs float64 primitives: {arch_name}, B={B}",
                   [(n, e, None, tol_of(n), "") for n, e in report])
    worst = sorted(report, key=lambda r: -r[1])[:8]
    print(f"\n{arch_name} B={B}: {len(report)} per-layer fp32 checks, largest relative errors: " +
          ", ".join(f"{n} {e:.2e}" for n, e in worst))
