"""Architectures the reference's API accepts beyond the 6-band / 3x3 default (VERDICT r3 "what's missing" #1, #2):

  * any band count: notebooks/training_example.ipynb:200-208 trains `train_deblender("des", None, ..., nb_of_bands=5)` on
    images[..., :5]; training/train.py:86,104-107 builds (59, 59, nb_of_bands);
  * kernels[i] other than 3: model/model.py:81-91,121-134 pass (kernels[i], kernels[i]) to Conv2D / Conv2DTranspose, and
    BASELINE's north_star names 3x3 / 5x5.

Oracle parity with the tolerances of tests/test_gpu_parity.py::_run_parity (outputs 2e-4 * max, ELBO 1e-4 relative,
gradients 1e-3 * max, Adam update), fp32 engine; band counts also on the bf16 engine with ITS tolerances
(tests/test_gpu_bf16.py::_run).  TF SAME padding of a 5x5 stride-2 layer is (2,2) on odd and (1,2) on even inputs, a
Conv2DTranspose mirrors it (oracle/vae_oracle.py::same_pad): the toy sizes 13 -> 7 -> 4 and 59 -> 30 -> 15 -> 8 -> 4 take
both cases.
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from tests.test_gpu_parity import _run_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bands", [5, 3, 1, 7, 8, 10, 15])
def test_band_counts_the_reference_api_accepts_toy_arch(bands):
    arch = vo.Arch(input_shape=(13, 13, bands), latent_dim=8, filters=(8, 16), kernels=(3, 3))
    _run_parity(arch, B=5, seed=40 + bands)
    _run_parity(arch, B=3, seed=50 + bands, train_decoder=False)


def test_five_bands_on_the_reference_architecture():
    # (59, 59, 5): the des / 5-band call of the reference's training notebook
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch(input_shape=(59, 59, 5))
    x, y = synthetic_stamps(8, seed=15, nb=5)
    _run_parity(arch, B=8, seed=61, data=(x, y))


def test_ten_bands_on_the_reference_architecture():
    # (59, 59, 10): train.py:86,104-107 builds (59, 59, nb_of_bands) for any band count; 8 .. 15 bands take the 16-channel
    # form of the folded first conv (bands + the constant 1 that carries the BatchNorm shift) and 32 head columns
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch(input_shape=(59, 59, 10))
    x, y = synthetic_stamps(8, seed=16, nb=10)
    _run_parity(arch, B=8, seed=62, data=(x, y))


@pytest.mark.parametrize("latent", [10, 5, 30, 1, 63])
def test_latent_sizes_that_are_not_multiples_of_four_toy_arch(latent):
    # model.py:164 takes any latent_dim: params_size = d + d (d + 1) / 2 is then rarely a multiple of 4; the engine pads the
    # rows of t / z / eps to 16-byte multiples and the two Dense kernels that touch them (zeros)
    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=latent, filters=(8, 16), kernels=(3, 3))
    _run_parity(arch, B=5, seed=140 + latent)
    _run_parity(arch, B=9, seed=150 + latent, train_decoder=False)


def test_latent_dim_ten_on_the_reference_architecture():
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch(latent_dim=10)
    x, y = synthetic_stamps(8, seed=17)
    _run_parity(arch, B=8, seed=63, data=(x, y))


def test_latent_dim_ten_through_the_api_sub_models():
    """encoder(x) -> (B, 65) rows, z(x).mean() / .stddev() -> (B, 10), decoder(z) and deblend over more stamps than one engine
    batch: the strided device rows are packed on every way out and in (dv_encode, dv_decode, dv_infer's mu / zstd / z)."""
    from debvader_amd.data import synthetic_stamps
    from debvader_amd.deblend_cutout.deblender import deblend
    from debvader_amd.model import model

    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=10, filters=(8, 16), kernels=(3, 3))
    net, encoder, decoder, z = model.create_model_vae((13, 13, 4), 10, [8, 16], [3, 3], max_batch=8, seed=4)
    eng = net._core.engine
    p = vo.init_params(arch, seed=11, perturb=0.05)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    eng.set_params(p)
    x, _ = synthetic_stamps(19, seed=18, size=13, nb=4)
    t = encoder(x).numpy()
    ref_t = vo.encoder_forward(arch, p, x.astype(np.float64), training=False)
    assert t.shape == (19, 65)
    np.testing.assert_allclose(t, ref_t, rtol=0, atol=2e-4 * np.abs(ref_t).max())
    q = z(x)
    np.testing.assert_array_equal(q.mean().numpy(), t[:, :10])
    L = vo.sampler_forward(arch, ref_t, np.zeros((19, 10)))[1]
    np.testing.assert_allclose(q.stddev().numpy(), np.sqrt((L ** 2).sum(-1)), rtol=2e-4)
    zz = np.random.default_rng(5).normal(size=(19, 10)).astype(np.float32)
    d = decoder(zz)
    loc, _ = vo.decoder_forward(arch, p, zz.astype(np.float64))
    np.testing.assert_allclose(d.mean().numpy(), loc, rtol=0, atol=2e-4 * np.abs(loc).max())
    mean, dist = deblend(net, x.astype(np.float64))
    assert mean.shape == (19, 13, 13, 4) and np.isfinite(mean).all() and (dist.stddev().numpy() >= 1e-4 * (1 - 1e-6)).all()
    eng.close()


def test_latent_dim_ten_on_the_bf16_engine():
    from tests.test_gpu_bf16 import _run

    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=10, filters=(16, 32), kernels=(3, 3))
    _run(arch, B=5, seed=75, check_fp64_grads=False)


@pytest.mark.parametrize("bands", [5, 3, 8, 10, 15])
def test_band_counts_on_the_bf16_engine(bands):
    """train.py:86,104-107 builds (59, 59, nb_of_bands) for any band count.  1 .. 7 bands: 16-column head tensors; 8 .. 15
    (round 6; the bf16 engine refused them until then): the normalised input still fits the 16-channel stamp-inner form
    (bands + the constant-one channel that carries beta), the folded first kernel's gradient has 16 input channels and the
    head tensors 32 columns (bf_head_kernel<32>, bf_colsum<32>, a 32-column head conv and 32-channel head gradient)."""
    from tests.test_gpu_bf16 import _run

    arch = vo.Arch(input_shape=(13, 13, bands), latent_dim=8, filters=(16, 32), kernels=(3, 3))
    # (against the bf16-rounding oracle and the fp64 oracle's outputs / ELBO; the loose per-tensor bound of the FORMAT's
    # cost on gradients is a property of the 4-band toy case of test_gpu_bf16.py, not of the band count)
    _run(arch, B=5, seed=70 + bands, check_fp64_grads=False)
    _run(arch, B=64, seed=80 + bands, tol_grad_b=5e-2, check_fp64_grads=False)
    if bands >= 8:
        _run(arch, B=256, seed=90 + bands, tol_grad_b=5e-2, check_fp64_grads=False)       # 256-stamp uniform tiles


def test_ten_bands_on_the_reference_architecture_bf16():
    """(59, 59, 10) on the bf16 engine at 64 stamps: the 32-column head forms at the reference's sizes, the dense trunk on
    the matrix cores, inference through deblend()'s entry point."""
    from tests.test_gpu_bf16 import _run

    arch = vo.Arch(input_shape=(59, 59, 10), latent_dim=32, filters=(32, 64, 128, 256), kernels=(3, 3, 3, 3))
    _run(arch, B=64, seed=97, tol_grad_b=0.25, min_cos=0.97, tol_grad_64=0.25)


@pytest.mark.parametrize("filters,kernels,size", [((16, 32), (5, 5), 13), ((16, 32, 64), (3, 5, 3), 20), ((32, 32), (1, 5), 13),
                                                  ((16, 32), (2, 4), 13), ((48, 80), (3, 3), 13), ((16, 48), (3, 5), 13)])
def test_kernel_sizes_other_than_three_on_the_bf16_engine(filters, kernels, size):
    """model.py:81-91,121-134 takes kernels[i] and filters[i] freely.  Filters that are multiples of 16 but not of 32 (48, 80)
    walk K in 8-channel pieces across tap boundaries in the conv tiles, and their weight gradients take the fp32 kernel as
    the k != 3 layers do.  bf16 engine: the one-pixel conv tiles walk a tap list of up to 25
    entries (forward and data gradient, both stride-2 forms); the weight gradient of such a layer runs on the fp32
    table-driven kernel over fp32 copies of its bf16 operands (exact).  Against the bf16-rounding oracle, for a ragged batch
    (general tiles), 64 stamps (64-stamp uniform tiles) and 256 stamps (256-stamp tiles, first-layer form)."""
    from tests.test_gpu_bf16 import _run

    arch = vo.Arch(input_shape=(size, size, 4), latent_dim=8, filters=filters, kernels=kernels)
    # (five stamps: a d(alpha) entry is a sum of five bf16-rounded products, so one flipped rounding - fp32 against float64
    # accumulation over up to 25 taps - moves it by up to 2^-8; 1e-2 instead of the 3 x 3 toy case's 5e-3)
    # (whole-step gradients of two bf16 evaluations differ by flipped roundings that the layers carry on: the bounds are
    # those of tests/test_gpu_bf16.py for nets of this depth; tests/test_gpu_0_layers_bf16.py checks every layer of a
    # k != 3 net ALONE at 1e-2 / 5e-3)
    tol = 0.25 if len(filters) > 2 else 5e-2
    _run(arch, B=5, seed=170 + kernels[0], tol_grad_b=tol, check_fp64_grads=False)
    _run(arch, B=64, seed=180 + kernels[0], tol_grad_b=tol, check_fp64_grads=False)
    _run(arch, B=256, seed=190 + kernels[0], tol_grad_b=tol, check_fp64_grads=False, train_decoder=False)


@pytest.mark.parametrize("kernels", [(5, 5), (5, 3), (1, 5), (2, 4)])
def test_kernel_sizes_other_than_three_toy_arch(kernels):
    arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(8, 16), kernels=kernels)
    _run_parity(arch, B=5, seed=90 + kernels[0] * 7 + kernels[1])
    _run_parity(arch, B=40, seed=95 + kernels[0] * 7 + kernels[1], train_decoder=False)


def test_five_by_five_kernels_on_the_reference_architecture():
    # (59, 59, 6) with kernels [5, 5, 5, 5]: K = 25 * Cin up to 6400, every SAME-pad case of the 59 -> 4 pyramid
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch(kernels=(5, 5, 5, 5))
    x, y = synthetic_stamps(6, seed=16)
    _run_parity(arch, B=6, seed=62, data=(x, y))


def test_bf16_engine_refuses_what_it_does_not_implement_with_a_message():
    from debvader_amd import engine as E
    from debvader_amd._lib import DvError

    with pytest.raises(DvError, match="filters must be multiples of 16"):
        E.Engine(E.make_config((13, 13, 4), 8, (8, 24), (3, 3), max_batch=4, dtype=1))
    with pytest.raises(DvError, match="bands"):                      # (1 .. 15 bands on both engines since round 6)
        E.Engine(E.make_config((13, 13, 16), 8, (16, 32), (3, 3), max_batch=4, dtype=1))


def test_train_deblender_with_five_bands_like_the_reference_notebook():
    """notebooks/training_example.ipynb:200-208: train_deblender("des", None, epochs, ..., nb_of_bands=5) on 5 + 5 stamps."""
    from debvader_amd.data import synthetic_stamps
    from debvader_amd.training.train import train_deblender

    x, y = synthetic_stamps(10, seed=3, nb=5)
    tr = (x[:5], y[:5])
    va = (x[5:], y[5:])
    hist_vae, hist_deb, net = train_deblender("des", None, 2, tr, va, tr, va, nb_of_bands=5, batch_size=5, verbose=0)
    for h in (hist_vae, hist_deb):
        assert set(h.history) >= {"loss", "mse", "kl_metric", "val_loss", "val_mse", "val_kl_metric"}
        assert len(h.history["loss"]) == 2 and np.isfinite(h.history["loss"]).all()
    mean = net(x[:3]).mean().numpy()
    assert mean.shape == (3, 59, 59, 5) and np.isfinite(mean).all()
