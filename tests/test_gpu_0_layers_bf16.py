"""Per-layer parity of the bf16 kernel family at the REAL layer shapes of the 59 x 59 x 6 net (model.py:79-98,112-137):
64^2 x 32, 32^2 x 64, 16^2 x 128, 8^2 x 256 (and the encoder's 59 / 30 / 15 / 8 / 4 grids), both strides, forward / data
gradient / weight gradient, with IDENTICAL bf16 operands on both sides so that nothing cascades.

Teacher forcing: one gradient step of the bf16 engine, then every layer is re-computed on its own by the oracle's
convolution primitives (oracle/vae_oracle.py: conv2d_same / convt2d_same and their gradients, float64) from the
ENGINE's stored input of that layer - the bf16 activation, the bf16 d(pre-activation) of the layer above, the fp32
master weights rounded to bf16 - and compared with what the engine stored for that layer.  What is left between the two
is the order of an fp32 sum and the bf16 rounding of the result (one ulp = 2^-8 of an element).
Stated tolerances: activations, pre-activations and d(pre-activations) <= 1e-2 * max; kernel gradients <= 5e-3 * max;
d(alpha) / d(bias) <= 1e-2 * max.  (Measured values are printed with -s.)

Batch 256 takes the 256-stamp tiles and, for the deep layers, the 128- / 64-stamp tiles (GT = 8 / 4) and the paired
32-channel chunks (CH = 2) of bconv_uni_kernel; batch 64 takes the 64-stamp forms; batch 48 (not a multiple of 64) the
general bconv_kernel and the separate PReLU backward.  The 128 x 128 x 6 / six-level architecture (BASELINE configs[3],
512-channel layers, no crop) runs at 16 stamps.
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from oracle import vae_oracle_bf16 as vb

pytestmark = pytest.mark.gpu

TOL_ACT, TOL_W, TOL_SMALL = 1e-2, 5e-3, 1e-2


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
    # kernel sizes other than 3 (model.py:81-91,121-134; round 5): tap lists of up to 25 entries in the conv tiles, the
    # weight gradient on the fp32 table-driven kernel over fp32 copies of the bf16 operands
    "59px-k5314": dict(kernels=(5, 3, 1, 4)),
    "29px-k55": dict(input_shape=(29, 29, 6), latent_dim=16, filters=(32, 64), kernels=(5, 5)),
    "29px-f48": dict(input_shape=(29, 29, 6), latent_dim=16, filters=(48, 80), kernels=(3, 5)),   # filters % 16 == 0 only
}


# (three ring stages are the non-default form since round 4: kept at 64 stamps and on the small 5 x 5 net, which cost seconds)
@pytest.mark.parametrize("arch_name,B,stages", [("59px", 256, 2), ("59px", 64, 2), ("59px", 64, 3),
                                                 ("59px", 48, 2), ("128px", 16, 2), ("59px-k5314", 64, 2),
                                                 ("29px-k55", 256, 2), ("29px-k55", 256, 3), ("29px-k55", 24, 3), ("29px-f48", 256, 2),
                                                 ("29px-f48", 40, 2)])
def test_every_conv_layer_alone_against_the_oracle_primitives(arch_name, B, stages, monkeypatch):
    from debvader_amd import engine as E
    from debvader_amd.data import synthetic_stamps

    # uniform conv tiles with two ring stages (the default since round 4) or three (DV_BCONV_NS2=0, read per launch)
    monkeypatch.setenv("DV_BCONV_NS2", "2" if stages == 2 else "0")

    arch = vo.Arch(**ARCHS[arch_name])
    L2 = 2 * len(arch.filters)
    p = vo.init_params(arch, seed=3, perturb=0.05)
    p["dec/head/bias"][arch.nb:] += 0.3          # sigma off its floor (see tests/test_gpu_bf16.py)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    x, y = synthetic_stamps(B, seed=11, size=arch.input_shape[0], nb=arch.nb)
    eps = np.random.default_rng(5).normal(size=(B, arch.latent_dim)).astype(np.float32)
    eng = E.Engine(E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels), max_batch=B,
                                 dtype=1))
    eng.set_params(p)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    eng.keep_outputs(True)          # (head_pre is read below: a step that keeps nothing runs the head inside the head conv)
    eng.grad_step(0, first=0, B=B, eps=eps)
    fused = ((B + 15) // 16 * 16) % 64 == 0      # PReLU backward inside the data-gradient epilogue (bconv_bwd_fusable)
    H, C = arch.input_shape[0], arch.nb
    report = []

    def act(name, shape):
        return eng.activation(name, shape).astype(np.float64)

    def enc_geom(j):
        lvl = j // 2
        hout = arch.enc_sizes[lvl + 1] if j % 2 else arch.enc_sizes[lvl]
        return hout, arch.filters[lvl], (2 if j % 2 else 1)

    def dec_geom(j):
        lvl = len(arch.filters) - 1 - j // 2
        hout = arch.w0 * 2 ** (j // 2 + 1)
        return hout, arch.filters[lvl], (2 if j % 2 == 0 else 1)

    # ---------------- encoder, forward ----------------
    xn = act("xn", (B, H, H, 16))
    assert np.abs(xn[..., C + 1:]).max() == 0.0
    xin0, w0 = vb._folded_first_conv(arch, p, xn[..., :C])
    assert np.array_equal(xin0[..., C], xn[..., C])          # the constant-one channel that carries beta
    enc_in = [None] * L2
    h = xin0
    for j in range(L2):
        hout, cout, s = enc_geom(j)
        w = w0 if j == 0 else vb.bf16(p[f"enc/conv{j}/kernel"])
        enc_in[j] = h
        u32 = vo.conv2d_same(h, w, p[f"enc/conv{j}/bias"], s)
        u = act(f"enc_u{j}", (B, hout, hout, cout))
        a = act(f"enc_a{j}", (B, hout, hout, cout))
        _check(report, f"enc_u{j}", u, vb.bf16(u32), TOL_ACT)
        _check(report, f"enc_a{j}", a, vb.bf16(vo.prelu(u32, p[f"enc/prelu{j}/alpha"])), TOL_ACT)
        h = a                                                 # teacher forcing: the ENGINE's activation feeds the next layer
    # ---------------- decoder, forward ----------------
    fl = arch.filters[-1]
    h = act("dec_in", (B, arch.w0, arch.w0, fl))
    dec_in = [None] * L2
    for j in range(L2):
        hout, cout, s = dec_geom(j)
        dec_in[j] = h
        u32 = vo.convt2d_same(h, vb.bf16(p[f"dec/convt{j}/kernel"]), p[f"dec/convt{j}/bias"], s)
        u = act(f"dec_u{j}", (B, hout, hout, cout))
        a = act(f"dec_a{j}", (B, hout, hout, cout))
        _check(report, f"dec_u{j}", u, vb.bf16(u32), TOL_ACT)
        _check(report, f"dec_a{j}", a, vb.bf16(vo.prelu(u32, p[f"dec/prelut{j}/alpha"])), TOL_ACT)
        h = a
    head_in = h
    wh = vb.bf16(p["dec/head/kernel"])
    tpre = vo.conv2d_same(head_in, wh, p["dec/head/bias"], 1)
    _check(report, "head_pre", act("head_pre", (B, arch.dec_out, arch.dec_out, 2 * C)), tpre, 2e-3)   # fp32 output

    # ---------------- backward: head, decoder ----------------
    dt = act("d_head_pre", (B, arch.dec_out, arch.dec_out, 16))
    assert np.abs(dt[..., 2 * C:]).max() == 0.0
    dt = dt[..., :2 * C]
    dh, dw, db = vo.conv2d_same_bwd(head_in, wh, dt, 1)
    _check(report, "dec/head/kernel", eng.get_grad("dec/head/kernel"), dw, TOL_W)
    _check(report, "dec/head/bias", eng.get_grad("dec/head/bias"), db, TOL_SMALL)
    for j in range(L2 - 1, -1, -1):
        hout, cout, s = dec_geom(j)
        u = act(f"dec_u{j}", (B, hout, hout, cout))
        alpha = p[f"dec/prelut{j}/alpha"]
        dA = dh if fused else vb.bf16(dh)
        du = act(f"dec_du{j}", (B, hout, hout, cout))
        _check(report, f"dec_du{j}", du, vb.bf16(dA * _gate(u, alpha)), TOL_ACT)
        _check(report, f"dec/prelut{j}/alpha", eng.get_grad(f"dec/prelut{j}/alpha"), (dA * np.minimum(u, 0)).sum(0), TOL_SMALL)
        _check(report, f"dec/convt{j}/bias", eng.get_grad(f"dec/convt{j}/bias"), (dA * _gate(u, alpha)).sum((0, 1, 2)), TOL_SMALL)
        dh, dk, _ = vo.convt2d_same_bwd(dec_in[j], vb.bf16(p[f"dec/convt{j}/kernel"]), du, s)   # from the ENGINE's du
        _check(report, f"dec/convt{j}/kernel", eng.get_grad(f"dec/convt{j}/kernel"), dk, TOL_W)
    trunk = vb.trunk_on_mfma(arch)
    if not trunk:
        _check(report, "d_dec_in", act("d_dec_in", (B, arch.w0, arch.w0, fl)), vb.bf16(dh), TOL_ACT)
    else:
        # ---------------- the dense trunk on the bf16 matrix cores (btrunk.hip), product by product ----------------
        # (teacher forcing as above: every product is recomputed from the ENGINE's own operands)
        bf = vb.bf16
        flat = arch.w0 * arch.w0 * fl
        a_last = act(f"enc_a{L2 - 1}", (B, arch.w0, arch.w0, fl)).reshape(B, flat)
        u_last = act(f"enc_u{L2 - 1}", (B, arch.w0, arch.w0, fl)).reshape(B, flat)
        al_flat, w_enc = p["enc/prelu_flat/alpha"], bf(p["enc/dense/kernel"])
        f_a = bf(vo.prelu(a_last, al_flat))
        tw = arch.params_size
        _check(report, "t", eng.activation("t", (B, tw)), f_a.dot(w_enc) + p["enc/dense/bias"], 2e-3)      # fp32 output
        ah = bf(eng.activation("dec_ah", (B, arch.dec_hidden)).astype(np.float64))
        w1 = bf(p["dec/dense1/kernel"])
        ur32 = ah.dot(w1) + p["dec/dense1/bias"]
        ur = act("dec_ur", (B, arch.w0, arch.w0, fl)).reshape(B, flat)
        _check(report, "dec_ur", ur, bf(ur32), TOL_ACT)
        _check(report, "dec_in", dec_in[0].reshape(B, flat), bf(vo.prelu(ur32, p["dec/prelu_r/alpha"])), TOL_ACT)
        # backward: the PReLU behind the trunk's Dense in the epilogue of the first transposed conv's data gradient
        dA = (dh if fused else bf(dh)).reshape(B, flat)
        al_r = p["dec/prelu_r/alpha"]
        du_r = act("d_dec_in", (B, arch.w0, arch.w0, fl)).reshape(B, flat)
        _check(report, "d_dec_ur", du_r, bf(dA * _gate(ur, al_r)), TOL_ACT)
        _check(report, "dec/prelu_r/alpha", eng.get_grad("dec/prelu_r/alpha"), (dA * np.minimum(ur, 0)).sum(0), TOL_SMALL)
        _check(report, "dec/dense1/bias", eng.get_grad("dec/dense1/bias"), (dA * _gate(ur, al_r)).sum(0), TOL_SMALL)
        _check(report, "dec/dense1/kernel", eng.get_grad("dec/dense1/kernel"), ah.T.dot(du_r), TOL_W)
        uh = eng.activation("dec_uh", (B, arch.dec_hidden)).astype(np.float64)
        _check(report, "d_dec_uh", eng.activation("d_dec_ah", (B, arch.dec_hidden)),
               du_r.dot(w1.T) * _gate(uh, p["dec/prelu_h/alpha"]), 2e-3)                                      # fp32 rows
        twp = (tw + 3) // 4 * 4
        d_t = bf(eng.activation("d_t", (B, twp)).astype(np.float64)[:, :tw])
        _check(report, "enc/dense/kernel", eng.get_grad("enc/dense/kernel"), f_a.T.dot(d_t), TOL_W)
        d_f = d_t.dot(w_enc.T)                                    # d(flatten PReLU output)
        dA7 = d_f * _gate(a_last, al_flat)
        al7 = p[f"enc/prelu{L2 - 1}/alpha"].reshape(flat)
        _check(report, f"enc_du{L2 - 1}", act(f"enc_du{L2 - 1}", (B, arch.w0, arch.w0, fl)).reshape(B, flat),
               bf(dA7 * _gate(u_last, al7)), TOL_ACT)
        _check(report, "enc/prelu_flat/alpha", eng.get_grad("enc/prelu_flat/alpha"), (d_f * np.minimum(a_last, 0)).sum(0), TOL_SMALL)
        _check(report, f"enc/prelu{L2 - 1}/alpha", eng.get_grad(f"enc/prelu{L2 - 1}/alpha").reshape(flat),
               (dA7 * np.minimum(u_last, 0)).sum(0), TOL_SMALL)
        _check(report, f"enc/conv{L2 - 1}/bias", eng.get_grad(f"enc/conv{L2 - 1}/bias"),
               (dA7 * _gate(u_last, al7)).reshape(B, -1, fl).sum((0, 1)), TOL_SMALL)

    # ---------------- backward: encoder (from the engine's d(pre-activation) of the last conv downwards) ----------------
    for j in range(L2 - 1, -1, -1):
        hout, cout, s = enc_geom(j)
        du = act(f"enc_du{j}", (B, hout, hout, cout))
        w = w0 if j == 0 else vb.bf16(p[f"enc/conv{j}/kernel"])
        dh, dw, _ = vo.conv2d_same_bwd(enc_in[j], w, du, s)
        if j > 0:
            _check(report, f"enc/conv{j}/kernel", eng.get_grad(f"enc/conv{j}/kernel"), dw, TOL_W)
            hp, cp, _ = enc_geom(j - 1)
            u = act(f"enc_u{j - 1}", (B, hp, hp, cp))
            alpha = p[f"enc/prelu{j - 1}/alpha"]
            dA = dh if fused else vb.bf16(dh)
            _check(report, f"enc_du{j - 1}", act(f"enc_du{j - 1}", (B, hp, hp, cp)), vb.bf16(dA * _gate(u, alpha)), TOL_ACT)
            _check(report, f"enc/prelu{j - 1}/alpha", eng.get_grad(f"enc/prelu{j - 1}/alpha"),
                   (dA * np.minimum(u, 0)).sum(0), TOL_SMALL)
            _check(report, f"enc/conv{j - 1}/bias", eng.get_grad(f"enc/conv{j - 1}/bias"),
                   (dA * _gate(u, alpha)).sum((0, 1, 2)), TOL_SMALL)
        else:
            W, gam, bet = p["enc/conv0/kernel"], p["enc/bn/gamma"], p["enc/bn/beta"]
            gk = dw[:, :, :C, :] * gam[None, None, :, None] + dw[:, :, C:C + 1, :] * bet[None, None, :, None]
            _check(report, "enc/conv0/kernel", eng.get_grad("enc/conv0/kernel"), gk, TOL_W)
            _check(report, "enc/bn/gamma", eng.get_grad("enc/bn/gamma"), (dw[:, :, :C, :] * W).sum((0, 1, 3)), TOL_SMALL)
            _check(report, "enc/bn/beta", eng.get_grad("enc/bn/beta"), (dw[:, :, C:C + 1, :] * W).sum((0, 1, 3)), TOL_SMALL)
    eng.close()
    worst = sorted(report, key=lambda r: -r[1])[:8]
    print(f"\n{arch_name} B={B}: {len(report)} per-layer checks, largest relative errors: " +
          ", ".join(f"{n} {e:.2e}" for n, e in worst))
