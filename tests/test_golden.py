"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py).

CPU: the oracle reproduces dc2_b4.npz (pins the oracle against regressions) and the package's helper
functions reproduce what the reference's own normalize.py / metrics.mse returned (helpers.npz).
GPU: the HIP engine reproduces dc2_b4.npz from the same inputs (real DC2 stamps of the reference's sample file).
"""
import os

import numpy as np
import pytest

from oracle import vae_oracle as vo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def summarise(a):
    a = np.asarray(a, np.float64).ravel()
    step = max(1, a.size // 64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), np.abs(a).max()], a[::step][:64]])


def _case():
    z = np.load(os.path.join(G, "dc2_b4.npz"))
    arch = vo.Arch()
    p = vo.init_params(arch, seed=int(z["param_seed"]), perturb=float(z["param_perturb"]))
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return z, arch, p


def test_oracle_reproduces_golden():
    z, arch, p = _case()
    x, y, eps = (z[k].astype(np.float64) for k in ("x", "y", "eps"))
    c = vo.forward(arch, p, x, eps, training=True)
    L = vo.losses(arch, c, y)
    g = vo.backward(arch, p, c, y)
    np.testing.assert_allclose(c["t"], z["t"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c["z"], z["z"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose([L["loss"], L["nll_mean"], L["kl_reg"], L["mse"]], z["loss"], rtol=1e-10)
    np.testing.assert_allclose(summarise(c["loc"]), z["loc_sum"], rtol=1e-9, atol=1e-12)
    for k, v in g.items():
        np.testing.assert_allclose(summarise(v), z["g/" + k], rtol=1e-7, atol=1e-9 * np.abs(z["g/" + k]).max(), err_msg=k)


def test_helpers_reproduce_reference_outputs():
    from debvader_amd.normalize.normalize import denormalize_non_linear, normalize_non_linear
    from debvader_amd.training.metrics import mse

    h = np.load(os.path.join(G, "helpers.npz"))
    np.testing.assert_array_equal(normalize_non_linear(h["a"]), h["normalized"])
    np.testing.assert_array_equal(denormalize_non_linear(h["normalized"]), h["denormalized"])
    assert mse(h["a"], h["b"]) == float(h["mse"])


@pytest.mark.gpu
def test_engine_reproduces_golden_on_real_dc2_stamps():
    from debvader_amd import engine as E

    z, arch, p = _case()
    eng = E.Engine(E.make_config(max_batch=4))
    eng.set_params(p)
    eng.optimizer_reset(1e-4)
    eng.upload(0, z["x"], z["y"])
    outf = eng.grad_step(0, first=0, B=4, eps=z["eps"])       # production form: the head kernel writes no loc / scale
    for i, k in enumerate(("loss", "nll_mean", "kl_reg", "mse")):
        assert abs(outf[k] - z["loss"][i]) <= 1e-4 * abs(z["loss"][i]), (k, outf[k], z["loss"][i])
    eng.keep_outputs(True)                                   # loc / scale are read back below
    out = eng.grad_step(0, first=0, B=4, eps=z["eps"])
    rel = lambda a, b: np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30)
    assert rel(eng.activation("t", (4, 560)), z["t"]) <= 2e-4
    assert rel(eng.activation("z", (4, 32)), z["z"]) <= 2e-4
    assert rel(eng.activation("kl", (4,)), z["kl"]) <= 2e-4
    for i, k in enumerate(("loss", "nll_mean", "kl_reg", "mse")):
        assert abs(out[k] - z["loss"][i]) <= 1e-4 * abs(z["loss"][i]), (k, out[k], z["loss"][i])   # ELBO tolerance
    assert rel(summarise(eng.activation("loc", (4, 59, 59, 6)))[4:], z["loc_sum"][4:]) <= 2e-4
    assert rel(summarise(eng.activation("scale", (4, 59, 59, 6)))[4:], z["scale_sum"][4:]) <= 2e-4
    for name, _, tr in arch.param_specs():
        if not tr:
            continue
        got, exp = summarise(eng.get_grad(name)), z["g/" + name]
        assert abs(got[2] - exp[2]) <= 1e-3 * exp[2], name                      # L2 norm
        assert np.abs(got[4:] - exp[4:]).max() <= 1e-3 * exp[3], name           # strided samples vs max|g|
    r = eng.infer(z["x"], eps=z["eps"], want=("loc", "scale"))
    assert rel(eng.encode(z["x"]), z["infer_t"]) <= 2e-4
    assert rel(summarise(r["loc"])[4:], z["infer_loc_sum"][4:]) <= 2e-4
    eng.close()
