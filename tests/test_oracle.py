"""Pins for oracle/vae_oracle.py (CPU only).

(i) structural pins that DO come from the reference: tensor shapes / parameter
totals of the shipped checkpoint + net.summary(), the fill_triangular known answer, the crop rule;
(ii) independent torch-CPU autograd cross-check; (iii) fp64 finite differences.
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo

torch = pytest.importorskip("torch")


def small_arch():
    # 2-level variant of the same plan: exercises odd sizes (11->6->3), the (0,1) pad and an odd crop
    return vo.Arch(input_shape=(11, 11, 3), latent_dim=4, filters=(4, 8), kernels=(3, 3))


def test_param_totals_match_reference_summary():
    # notebooks/deblender_to_onnx.ipynb:158,225,229-231
    a = vo.Arch()
    enc, dec = a.param_counts()
    assert enc == 3_741_224
    assert dec == 4_577_228
    assert enc + dec == 8_318_452
    specs = a.param_specs()
    assert len(specs) == 64
    trainable = sum(int(np.prod(s)) for _, s, t in specs if t)
    assert trainable == 8_318_440
    enc_train = sum(int(np.prod(s)) for n, s, t in specs if t and n.startswith("enc/"))
    assert enc_train == 3_741_212         # decoder frozen: 3 741 212 trainable / 4 577 240 not
    assert a.params_size == 560 and a.dec_hidden == 560 and a.flat == 4096 and a.w0 == 4


def test_checkpoint_index_shapes():
    # SURVEY 8(a) table (parsed from weights_noisy_v4.386--6.61.ckpt.index)
    s = {n: sh for n, sh, _ in vo.Arch().param_specs()}
    assert s["enc/conv0/kernel"] == (3, 3, 6, 32) and s["enc/prelu0/alpha"] == (59, 59, 32)
    assert s["enc/prelu1/alpha"] == (30, 30, 32) and s["enc/prelu3/alpha"] == (15, 15, 64)
    assert s["enc/prelu5/alpha"] == (8, 8, 128) and s["enc/prelu7/alpha"] == (4, 4, 256)
    assert s["enc/dense/kernel"] == (4096, 560)
    assert s["dec/prelu_in/alpha"] == (32,) and s["dec/dense0/kernel"] == (32, 560)
    assert s["dec/dense1/kernel"] == (560, 4096)
    assert s["dec/convt0/kernel"] == (3, 3, 256, 256) and s["dec/prelut0/alpha"] == (8, 8, 256)
    assert s["dec/convt2/kernel"] == (3, 3, 128, 256) and s["dec/convt4/kernel"] == (3, 3, 64, 128)
    assert s["dec/convt6/kernel"] == (3, 3, 32, 64) and s["dec/prelut7/alpha"] == (64, 64, 32)
    assert s["dec/head/kernel"] == (3, 3, 32, 12)


def test_fill_triangular_known_answer():
    got = vo.fill_triangular(np.arange(1.0, 7.0))
    np.testing.assert_array_equal(got, [[4, 0, 0], [6, 5, 0], [3, 2, 1]])
    # split sizes printed in the saved net.summary(): 528 -> concat(496, 528)=1024 -> (32,32)
    assert vo.fill_triangular(np.zeros(528)).shape == (32, 32)


def test_same_padding_and_crop_rules():
    assert vo.same_pad(59, 3, 2) == (30, 1, 1)
    assert vo.same_pad(30, 3, 2) == (15, 0, 1)
    assert vo.same_pad(15, 3, 2) == (8, 1, 1)
    assert vo.same_pad(8, 3, 2) == (4, 0, 1)
    assert vo.same_pad(59, 3, 1) == (59, 1, 1)
    a = vo.Arch()
    assert a.dec_out == 64 and a.crop == (2, 3)      # model.py:146-148
    assert a.enc_sizes == [59, 30, 15, 8, 4]


def _to_torch(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


@pytest.mark.parametrize("arch_fn,B", [(small_arch, 3), (vo.Arch, 2)])
def test_forward_backward_vs_torch_autograd(arch_fn, B):
    from tests import torch_ref

    arch = arch_fn()
    rng = np.random.default_rng(1)
    p = vo.init_params(arch, seed=3, perturb=0.05)
    H, W, C = arch.input_shape
    x = rng.normal(0, 0.3, size=(B, H, W, C))
    y = np.abs(rng.normal(0, 0.3, size=(B, H, W, C)))
    eps = rng.normal(size=(B, arch.latent_dim))
    c = vo.forward(arch, p, x, eps, training=True)
    out = vo.losses(arch, c, y)
    g = vo.backward(arch, p, c, y)

    pt = _to_torch(p)
    r = torch_ref.net_loss(arch, pt, torch.tensor(x), torch.tensor(y), torch.tensor(eps), training=True)
    r["loss"].backward()
    np.testing.assert_allclose(c["t"], r["t"].detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(c["z"], r["z"].detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(c["kl"], r["kl"].detach().numpy(), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(c["loc"], r["loc"].detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(c["scale"], r["scale"].detach().numpy(), rtol=1e-9, atol=1e-11)
    assert abs(out["loss"] - r["loss"].item()) <= 1e-10 * abs(out["loss"])
    assert abs(out["kl_reg"] - r["kl_reg"].item()) <= 1e-8 * abs(out["kl_reg"]) + 1e-14
    for name, _, tr in arch.param_specs():
        if not tr:
            continue
        gt = pt[name].grad.numpy()
        scale = np.abs(gt).max() + 1e-30
        np.testing.assert_allclose(g[name] / scale, gt / scale, rtol=0, atol=2e-8, err_msg=name)


def test_inference_mode_uses_moving_stats():
    from tests import torch_ref

    arch = small_arch()
    rng = np.random.default_rng(5)
    p = vo.init_params(arch, seed=2, perturb=0.1)
    x = rng.normal(size=(2, 11, 11, 3))
    eps = rng.normal(size=(2, 4))
    c = vo.forward(arch, p, x, eps, training=False)
    pt = _to_torch(p)
    r = torch_ref.net_loss(arch, pt, torch.tensor(x), torch.tensor(np.abs(x)), torch.tensor(eps), training=False)
    np.testing.assert_allclose(c["loc"], r["loc"].detach().numpy(), rtol=1e-9, atol=1e-11)


def test_finite_difference_gradients():
    arch = small_arch()
    arch.sigma_floor = 0.05      # keep 1/sigma^2 curvature small enough for central differences
    rng = np.random.default_rng(7)
    p = vo.init_params(arch, seed=11, perturb=0.05)
    x = rng.normal(0, 0.5, size=(2, 11, 11, 3))
    y = np.abs(rng.normal(0, 0.5, size=(2, 11, 11, 3)))
    eps = rng.normal(size=(2, 4))

    def f(pp):
        c = vo.forward(arch, pp, x, eps, training=True)
        return vo.losses(arch, c, y)["loss"]

    c = vo.forward(arch, p, x, eps, training=True)
    g = vo.backward(arch, p, c, y)
    for name in ["enc/conv1/kernel", "enc/prelu0/alpha", "enc/dense/kernel", "enc/bn/gamma",
                 "dec/convt0/kernel", "dec/convt1/kernel", "dec/prelut2/alpha", "dec/head/bias", "dec/dense0/bias"]:
        flat_idx = rng.integers(0, p[name].size, size=3)
        for fi in flat_idx:
            idx = np.unravel_index(fi, p[name].shape)
            h = 1e-6
            pp = {k: v.copy() for k, v in p.items()}
            pp[name][idx] += h
            fp = f(pp)
            pp[name][idx] -= 2 * h
            fm = f(pp)
            num = (fp - fm) / (2 * h)
            assert abs(num - g[name][idx]) <= 1e-5 * max(1.0, abs(num)), (name, idx, num, g[name][idx])


def test_data_parallel_shard_sum_equals_full_batch():
    """with global normalisers, per-shard losses/grads SUM to the full-batch ones
    (BN statistics supplied globally)."""
    arch = small_arch()
    rng = np.random.default_rng(9)
    p = vo.init_params(arch, seed=4, perturb=0.05)
    x = rng.normal(size=(4, 11, 11, 3))
    y = np.abs(rng.normal(size=(4, 11, 11, 3)))
    eps = rng.normal(size=(4, 4))
    # make BN shard-independent by running in inference-statistics mode with the global batch stats
    p["enc/bn/moving_mean"] = x.mean(axis=(0, 1, 2))
    p["enc/bn/moving_variance"] = x.var(axis=(0, 1, 2))
    cf = vo.forward(arch, p, x, eps, training=False)
    full = vo.losses(arch, cf, y)
    gf = vo.backward(arch, p, cf, y)
    tot, gs = 0.0, None
    for sl in (slice(0, 2), slice(2, 4)):
        c = vo.forward(arch, p, x[sl], eps[sl], training=False)
        tot += vo.losses(arch, c, y[sl], global_batch=4)["loss"]
        g = vo.backward(arch, p, c, y[sl], global_batch=4)
        gs = g if gs is None else {k: gs[k] + g[k] for k in g}
    assert abs(tot - full["loss"]) < 1e-12 * abs(full["loss"])
    for k in gf:
        np.testing.assert_allclose(gs[k], gf[k], rtol=1e-9, atol=1e-13, err_msg=k)


def test_adam_matches_closed_form_first_step():
    st = vo.AdamState()
    p = {"w": np.array([1.0, -2.0])}
    g = {"w": np.array([0.5, -0.25])}
    vo.adam_step(st, p, g)
    # first legacy-Adam step: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2
    lr_t = 1e-4 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([1.0, -2.0]) - lr_t * (0.1 * g["w"]) / (np.sqrt(0.001 * g["w"] ** 2) + 1e-7)
    np.testing.assert_allclose(p["w"], exp, rtol=1e-14)


def test_philox_known_answer():
    # Random123 KAT: counter=0,key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    r = vo.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32))
    assert [hex(v) for v in r[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    r = vo.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full((1, 2), 0xFFFFFFFF, np.uint32))
    assert [hex(v) for v in r[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    e = vo.philox_normal(123, 0, 4096, 32)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01
