"""CPU checks of the bf16-storage oracle (oracle/vae_oracle_bf16.py): with the rounding switched off it must reproduce
vae_oracle (this pins the folded first conv and its unfolding), and with it on it must stay within bf16's reach."""
import numpy as np

from oracle import vae_oracle as vo
from oracle import vae_oracle_bf16 as vb


def _case(arch, B, seed):
    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    H, W, C = arch.input_shape
    x = rng.normal(0, 0.4, size=(B, H, W, C))
    y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C)))
    eps = rng.normal(size=(B, arch.latent_dim))
    return p, x, y, eps


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -9, -0.1, 3.0e-41, 65504.0], np.float64)
    r = vb.bf16(x)
    assert r[0] == 1.0
    assert r[1] == 1.0                      # tie -> even mantissa
    assert r[2] == 1.0 + 2 ** -7            # tie -> even (upwards)
    assert abs(r[3] + 0.1) < 0.1 * 2 ** -8
    bits = np.ascontiguousarray(r, np.float32).view(np.uint32)
    assert np.all((bits & 0xFFFF) == 0)


def test_without_rounding_it_is_the_fp_oracle(monkeypatch):
    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3))
    p, x, y, eps = _case(arch, 3, 5)
    monkeypatch.setattr(vb, "bf16", lambda a: np.asarray(a))
    c = vb.forward(arch, p, x, eps, training=True)
    c0 = vo.forward(arch, p, x, eps, training=True)
    for k in ("t", "z", "kl", "loc", "scale", "head_pre"):
        np.testing.assert_allclose(c[k], c0[k], rtol=1e-7, atol=1e-10)
    for fused in (True, False):
        g = vb.backward(arch, p, c, y, fused=fused)
        g0 = vo.backward(arch, p, c0, y)
        assert set(g) == set(g0)
        for k in g0:
            np.testing.assert_allclose(g[k], g0[k], rtol=1e-6, atol=1e-10, err_msg=k)


def test_with_rounding_it_stays_close():
    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3))
    p, x, y, eps = _case(arch, 3, 6)
    c = vb.forward(arch, p, x, eps, training=True)
    c0 = vo.forward(arch, p, x, eps, training=True)
    assert np.abs(c["t"] - c0["t"]).max() <= 3e-2 * np.abs(c0["t"]).max()
    assert np.abs(c["head_pre"] - c0["head_pre"]).max() <= 3e-2 * np.abs(c0["head_pre"]).max()
