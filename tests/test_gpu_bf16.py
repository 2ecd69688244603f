"""GPU parity of the bf16 engine (BASELINE configs[2]: bf16 storage / bf16 MFMA operands, fp32 accumulation, fp32 master
weights, fp32 dense trunk / sampler / head) through the C-ABI.

Two references, two kinds of tolerance:
  * oracle/vae_oracle_bf16.py rounds to bfloat16 exactly where the engine does.  Against it the engine must agree as
    two orderings of the same fp32/fp64 sums do - every residual difference is a bf16 rounding that flipped (one ulp =
    2^-8) and what it seeds downstream.  Stated: outputs <= 1e-2 * max, ELBO scalars <= 5e-4 relative, gradients
    <= 5e-3 * max per tensor where no rounding flips (toy net, small batch), <= 5e-2 * max with the fused epilogues on
    a 64-stamp batch.
  * oracle/vae_oracle.py (float64 restatement of the reference, model.py:61-161) measures what the FORMAT costs.
    Stated: outputs <= 2e-2 * max, ELBO scalars <= 5e-3 relative, gradients: cosine >= 0.97 per tensor and
    <= 0.25 * max on the 59 x 59 x 6 net.  (The engine and the bf16 oracle sit at the same distance from float64:
    tools/bf16_probe.py prints all three.)
The head's sigma is kept off its 1e-4 floor (bias + 0.3 on the scale channels): at the floor 1/sigma^2 = 1e8 turns a
one-ulp bf16 change of the mean into an O(1) change of the gradient and no two implementations agree - a property of
the loss (model.py:154-159), see DESIGN.md.
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from oracle import vae_oracle_bf16 as vb
from tests import margins

pytestmark = pytest.mark.gpu


def _relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _cos(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a.dot(b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _case(arch, B, seed, data=None):
    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    H, W, C = arch.input_shape
    if data is None:
        x = rng.normal(0, 0.4, size=(B, H, W, C)).astype(np.float32)
        y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C))).astype(np.float32)
    else:
        x, y = data
    eps = rng.normal(size=(B, arch.latent_dim)).astype(np.float32)
    p["dec/head/bias"][arch.nb:] += 0.3
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return p, x, y, eps


def _engine(arch, B, dtype=1):
    from debvader_amd import engine as E

    return E.Engine(E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels),
                                  max_batch=B, dtype=dtype))


def toy_arch():
    # 13 -> 7 -> 4: both SAME-pad cases, odd crop; 16 / 32 filters are the narrowest the bf16 kernels take
    return vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3))


def _run(arch, B, seed, data=None, train_decoder=True, tol_out=1e-2, tol_grad_b=5e-3, check_fp64_grads=True,
         min_cos=0.97, tol_grad_64=0.25, tol_scal_64=5e-3, data_seed=None, per_tensor=False):
    from tests import oracle_jobs, oracle_pool

    if data_seed is not None:
        data = oracle_jobs.stamps(B, data_seed)
    p, x, y, eps = _case(arch, B, seed, data)
    eng = _engine(arch, B)
    eng.set_params(p)
    eng.set_trainable(True, train_decoder)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    eng.keep_outputs(True)
    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
    # both oracles (bf16-rounding and float64); seed-built cases may come from a worker process (tests/oracle_pool.py)
    if (data is None or data_seed is not None) and arch == oracle_jobs.make_arch(oracle_jobs.arch_kw(arch)):
        ev = oracle_pool.fetch("bf16_case", arch_kw=oracle_jobs.arch_kw(arch), B=B, seed=seed, data_seed=data_seed,
                               train_decoder=train_decoder)
    else:
        ev = oracle_jobs.bf16_eval(arch, p, x, y, eps, train_decoder)
    cb, rb, gb, c, r, g = ev["cb"], ev["rb"], ev["gb"], ev["c"], ev["r"], ev["g"]

    out = eng.grad_step(0, first=0, B=B, eps=eps)
    H, W, C = arch.input_shape
    d = arch.latent_dim
    shapes = {"t": (B, arch.params_size), "z": (B, d), "kl": (B,), "loc": (B, H, W, C), "scale": (B, H, W, C),
              "head_pre": (B, arch.dec_out, arch.dec_out, 2 * C)}
    for k, shape in shapes.items():
        v = eng.activation(k, shape)
        assert _relmax(v, cb[k]) <= tol_out, ("bf16 oracle", k, _relmax(v, cb[k]))
        assert _relmax(v, c[k]) <= 2e-2, ("fp64 oracle", k, _relmax(v, c[k]))
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        assert abs(out[k] - rb[k]) <= 5e-4 * abs(rb[k]) + 1e-9, ("bf16 oracle", k, out[k], rb[k])
        assert abs(out[k] - r[k]) <= tol_scal_64 * abs(r[k]) + 1e-9, ("fp64 oracle", k, out[k], r[k])
    assert set(gb) == set(g)
    rows = [(k, abs(out[k] - rb[k]) / (abs(rb[k]) + 1e-30), abs(out[k] - r[k]) / (abs(r[k]) + 1e-30), 5e-4,
             f"ELBO scalar, relative; other = vs float64 oracle (bound {tol_scal_64:g})") for k in ("loss", "nll_mean", "kl_reg")]
    failed = []
    for name in g:
        gg = eng.get_grad(name)
        eb, ef, cs = _relmax(gg, gb[name]), _relmax(gg, g[name]), _cos(gg, g[name])
        rows.append((name, eb, ef, tol_grad_b, f"gradient; other = vs float64 oracle (bound {tol_grad_64:g}), cosine {cs:.4f} "
                                                f"(bound {min_cos:g}); bf16 oracle vs float64: {_relmax(gb[name], g[name]):.3e}"))
        # against the bf16-rounding oracle: the flat bound, and - per_tensor: the whole-step cases of the large nets at
        # their quoted batches, VERDICT r5 item 4(iii) - per tensor no more than twice what the bf16 oracle itself is away
        # from float64 (two orderings of the same rounded sums must not differ by more than the format costs), floor 1e-2
        bound_b = min(tol_grad_b, max(2.0 * _relmax(gb[name], g[name]), 1e-2)) if per_tensor else tol_grad_b
        rows[-1] = (name, eb, ef, bound_b, rows[-1][4])
        if eb > bound_b:
            failed.append(("bf16 oracle", name, eb, bound_b))
        if check_fp64_grads and (cs < min_cos or ef > tol_grad_64):
            failed.append(("fp64 oracle", name, ef, cs))
    margins.record(f"bf16 engine vs bf16-rounding oracle and float64 oracle: {'x'.join(map(str, arch.input_shape))}, "
                   f"{len(arch.filters)} levels, B={B}, {'stage 1' if train_decoder else 'decoder frozen'}, head scale bias +0.3",
                   rows, "engine = vs the bf16-rounding oracle (same storage roundings); errors are max|a - ref| / max|ref| per tensor")
    assert not failed, failed
    if not train_decoder:
        for name, _, tr in arch.param_specs():
            if name.startswith("dec/"):
                assert name not in g

    # one train step: legacy Adam on the fp32 master weights with the engine's own gradients, then the bf16 matrices
    # are re-cast (a second forward must see the new weights)
    gg = {name: eng.get_grad(name).astype(np.float64) for name in g}
    st = vo.AdamState()
    p2 = {k: v.copy() for k, v in p.items()}
    vo.adam_step(st, p2, gg)
    out2 = eng.train_step(0, first=0, B=B, eps=eps)
    assert abs(out2["loss"] - out["loss"]) <= 1e-5 * abs(out["loss"]) + 1e-9      # same forward, bit-level noise only
    for name in g:
        got = eng.get_param(name)
        assert np.abs(got - p2[name]).max() <= 1e-6, (name, np.abs(got - p2[name]).max())
    out3 = eng.eval_step(0, first=0, B=B, eps=eps)
    c3 = vb.forward(arch, {k: eng.get_param(k).astype(np.float64) for k in p}, x64, e64, training=False)
    r3 = vo.losses(arch, c3, y64)
    assert abs(out3["loss"] - r3["loss"]) <= 2e-3 * abs(r3["loss"]) + 1e-6, (out3["loss"], r3["loss"])
    eng.close()


def test_toy_arch_small_batch_unfused_backward():
    _run(toy_arch(), B=5, seed=0)


def test_toy_arch_frozen_decoder():
    _run(toy_arch(), B=3, seed=4, train_decoder=False)


def test_toy_arch_64_stamps_fused_epilogues():
    _run(toy_arch(), B=64, seed=1, tol_grad_b=5e-2)


def test_toy_arch_ragged_batch_with_fused_epilogues():
    # 50 stamps pad to 64: the pad rows must contribute nothing to any batch reduction
    _run(toy_arch(), B=50, seed=2, tol_grad_b=5e-2)


def test_toy_arch_256_and_512_stamps_uniform_tile_kernel():
    """Stamps padded to a multiple of 256 take bconv_uni_kernel (one output pixel x 256 stamps per tile, scalar tap
    bookkeeping, accumulators preloaded with the bias): one and two tiles per pixel."""
    _run(toy_arch(), B=256, seed=7, tol_grad_b=5e-2)
    _run(toy_arch(), B=500, seed=8, tol_grad_b=5e-2)


def test_wide_channel_arch_all_column_tile_widths():
    """Three levels up to 128 channels on 20 x 20 stamps: K loops of several 32-channel chunks per tap, 64-wide column
    tiles (DV_BCONV_MIN_TILES=1 keeps the launcher from narrowing them for these few-pixel layers) and the narrowed
    ones, in the uniform-tile kernel (256 stamps) and in the general one (48 stamps: a tile spans several pixels)."""
    import os

    arch = vo.Arch(input_shape=(20, 20, 4), latent_dim=8, filters=(32, 64, 128), kernels=(3, 3, 3))
    # (d(gamma) / d(beta) of the input BatchNorm are four numbers each, sums over every pixel with heavy
    # cancellation: they carry the loosest agreement, 0.1-0.2 of their maximum between any two bf16 evaluations)
    # (outputs at 1.5e-2 * max, round 6: with the dense trunk on the matrix cores one more chain of bf16 roundings
    # sits between the two orderings of the same sums - measured 1.07e-2 on `loc` at 256 stamps, one flipped
    # rounding of the hidden layer; every product of the trunk alone stays within 1e-2: test_gpu_0_layers_bf16.py)
    for B, seed in ((256, 11), (48, 12)):                # (each case under both tile rules in a row: one oracle evaluation)
        for min_tiles in ("1", None):
            if min_tiles is None:
                os.environ.pop("DV_BCONV_MIN_TILES", None)
            else:
                os.environ["DV_BCONV_MIN_TILES"] = min_tiles
            try:
                _run(arch, B=B, seed=seed, tol_out=1.5e-2, tol_grad_b=0.25, min_cos=0.95, tol_grad_64=0.35)
            finally:
                os.environ.pop("DV_BCONV_MIN_TILES", None)


def test_full_arch_dc2_stamps():
    """The reference's 59 x 59 x 6 / [32,64,128,256] net on its own sample stamps (tests/golden, the first 4 DC2 stamps)."""
    import os

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "dc2_b4.npz"))
    x = gold["x"].astype(np.float32)
    y = gold["y"].astype(np.float32)
    # rounding flips cascade through 17 layers: against the bf16 oracle the gradients agree as well as the bf16 oracle
    # agrees with float64
    _run(vo.Arch(), B=4, seed=2, data=(x, y), tol_grad_b=0.4, min_cos=0.9, tol_grad_64=0.5)


def test_full_arch_64_stamps():
    from debvader_amd.data import synthetic_stamps

    _run(vo.Arch(), B=64, seed=3, data_seed=6, tol_grad_b=0.25, min_cos=0.97, tol_grad_64=0.25, per_tensor=True)


def test_full_arch_at_the_quoted_batch_of_256_stamps():
    """BASELINE configs[2]'s per-GPU batch (VERDICT r4 "what's weak" #3): the WHOLE bf16 step of the 59 x 59 x 6 net at 256
    stamps - the 256-stamp uniform tiles, the 128- / 64-stamp forms of the deep layers, paired chunks, the fused
    PReLU-backward epilogues - against the bf16-rounding oracle (ELBO scalars <= 5e-4 relative, outputs <= 1e-2 * max) and
    the float64 oracle (ELBO <= 5e-3 relative, outputs <= 2e-2 * max).  Gradients per tensor: <= 0.25 * max against the
    bf16 oracle and against float64, cosine >= 0.97 - the bounds of the 64-stamp case; every flipped bf16 rounding of an
    activation seeds a difference that 17 layers carry on, and the two bf16 evaluations sit as far from each other as
    each sits from float64 (the measured numbers per tensor: profiles/r05_parity_margins.txt)."""
    _run(vo.Arch(), B=256, seed=5, data_seed=9, tol_grad_b=0.25, min_cos=0.97, tol_grad_64=0.25, per_tensor=True)


def test_inference_matches_the_bf16_oracle_and_other_entry_points():
    arch = toy_arch()
    B = 37
    p, x, y, eps = _case(arch, B, 9)
    eng = _engine(arch, 64)
    eng.set_params(p)
    res = eng.infer(x, eps=eps, want=("loc", "scale", "mu", "z"))
    cb = vb.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=False)
    assert _relmax(res["loc"], cb["loc"]) <= 1e-2
    assert _relmax(res["scale"], cb["scale"]) <= 1e-2
    assert _relmax(res["z"], cb["z"]) <= 1e-2
    t = eng.encode(x)
    assert _relmax(t, cb["t"]) <= 1e-2
    loc, scale = eng.decode(cb["z"].astype(np.float32))
    assert _relmax(loc, cb["loc"]) <= 1e-2
    eng.close()


def test_bf16_rejects_unsupported_filters():
    from debvader_amd import engine as E
    from debvader_amd._lib import DvError

    with pytest.raises(DvError):
        E.Engine(E.make_config((13, 13, 4), 8, (8, 16), (3, 3), max_batch=4, dtype=1))


def _training_run(kind, steps, sigma_floor, shift, data, val, B=64, lr=1e-4):
    """`steps` legacy-Adam steps (train.py:104-107,125-130) of one engine from a fixed initialisation on fixed batches.
    kind: "f32" (the fp32 engine as shipped), "f32alt" (the fp32 engine with the general gather-GEMM / tiled
    weight-gradient kernels instead of the strip and fused stride-2 forms: the SAME arithmetic in another summation
    order) or "bf16".  Returns per-step losses, per-step count of pixels on the sigma floor, final validation loss."""
    from debvader_amd import engine as E
    from debvader_amd._lib import check
    from tests import debug_lib
    import contextlib

    x, y = data
    xv, yv = val
    # "f32alt" runs on the development build of the library, whose process-wide switch selects the other kernel family
    with (debug_lib.debug_build() if kind == "f32alt" else contextlib.nullcontext()) as dlib:
        if dlib is not None:
            check(dlib.dv_debug_general_kernels(1))
        try:
            eng = E.Engine(E.make_config(max_batch=B, dtype=1 if kind == "bf16" else 0, sigma_floor=sigma_floor))
            eng.init(seed=5)
            if shift:
                hb = eng.get_param("dec/head/bias")
                hb[6:] += shift
                eng.set_param("dec/head/bias", hb)
            eng.optimizer_reset(lr)
            eng.upload(0, x, y)
            eng.upload(1, xv, yv)
            eng.keep_outputs(True)
            nb = x.shape[0] // B
            losses, floor_px = [], []
            for s in range(steps):
                out = eng.train_step(0, first=(s % nb) * B, B=B, seed=100 + s)
                losses.append(out["loss"])
                floor_px.append(int((eng.activation("scale", (B, 59, 59, 6)) <= sigma_floor * (1 + 1e-5)).sum()))
            v = np.mean([eng.eval_step(1, first=k * B, B=B, seed=7000 + k)["loss"] for k in range(xv.shape[0] // B)])
            eng.close()
        finally:
            if dlib is not None:
                check(dlib.dv_debug_general_kernels(0))
    return np.asarray(losses, np.float64), np.asarray(floor_px), float(v)


@pytest.fixture(scope="module")
def training_data():
    from debvader_amd.data import synthetic_stamps

    x, y = synthetic_stamps(512 + 128, seed=21)
    return (x[:512], y[:512]), (x[512:], y[512:])


def test_bf16_training_quality_over_200_steps_against_two_fp32_summation_orders(training_data):
    """What the bf16 FORMAT costs a training run, measured where a measurement is possible: 200 Adam steps of the three
    engines configurations from the same initialisation on the same batches, with the head's sigma floor at 0.05
    instead of the reference's 1e-4 (model.py:154-159; the floor is a dv_config field).  With the floor at 1e-4 the loss
    is chaotic - see the next test - and no two runs can be compared; at 0.05 its curvature is bounded (1 / sigma^2 <=
    400), the fp32 engine under another summation order tracks itself to a few 1e-2 per step and 1e-3 in the final
    validation loss, and the bf16 engine must stay within a stated band of both:
    per step |bf16 - f32| <= 0.1 (measured 0.05; the loss falls from +2.2e3 to -1.9 over the run), median <= 5e-3
    (measured 1.7e-3), final validation loss within 1e-2 (measured 2.5e-3 = 0.13 %; the fp32 control: 6e-4)."""
    data, val = training_data
    runs = {k: _training_run(k, 200, 0.05, 0.0, data, val) for k in ("f32", "f32alt", "bf16")}
    f32, falt, bf = (runs[k][0] for k in ("f32", "f32alt", "bf16"))
    assert np.all(np.isfinite(bf))
    assert f32[-20:].mean() < -1.5 < 0.0 < f32[:3].mean()         # the run trains (the loss starts at ~2e3)
    late = slice(20, None)                                         # (the first steps fall through three decades)
    d_ctl, d_bf = np.abs(falt - f32)[late], np.abs(bf - f32)[late]
    print(f"\nper-step |f32alt-f32| max {d_ctl.max():.4f} median {np.median(d_ctl):.5f}; |bf16-f32| max {d_bf.max():.4f} "
          f"median {np.median(d_bf):.5f}; validation f32 {runs['f32'][2]:.5f} f32alt {runs['f32alt'][2]:.5f} "
          f"bf16 {runs['bf16'][2]:.5f}")
    assert d_ctl.max() <= 0.1 and abs(runs["f32alt"][2] - runs["f32"][2]) <= 5e-3            # the control holds
    assert d_bf.max() <= 0.1 and np.median(d_bf) <= 5e-3
    assert abs(runs["bf16"][2] - runs["f32"][2]) <= 1e-2


def test_at_the_reference_sigma_floor_two_fp32_summation_orders_separate_like_bf16_does(training_data):
    """Round 2's 40-step comparison of the bf16 and the fp32 engine failed after ~20 steps; this is the evidence for why
    (tools/bf16_drift.py, profiles/r03_bf16_drift_*.txt).  With the reference's floor of 1e-4 under sigma the first
    pixel reaches the floor at the SAME step in every run (a property of the data and the optimiser, not of a kernel);
    there (y - mu)^2 / sigma^2 is ~1e8 (y - mu)^2, the loss jumps by orders of magnitude for a step and Adam turns the
    spike into a full-size update, after which any two runs decorrelate - the fp32 engine under a different SUMMATION
    ORDER just as much as the bf16 engine.  Asserted: (i) until that step the fp32 control agrees with fp32 to 1e-3 of
    the loss scale and bf16 to 2e-2; (ii) the first floor pixel arrives at the same step in all three runs; (iii) past
    it the fp32 control is further from fp32 than 20x its distance before (it has separated, with identical arithmetic)
    and the bf16 run is no further from fp32 than 4x the control is (median over the remaining steps)."""
    data, val = training_data
    steps = 120
    runs = {k: _training_run(k, steps, 1e-4, 0.3, data, val) for k in ("f32", "f32alt", "bf16")}
    f32, falt, bf = (runs[k][0] for k in ("f32", "f32alt", "bf16"))
    first = {k: int(np.argmax(runs[k][1] > 0)) if (runs[k][1] > 0).any() else -1 for k in runs}
    print(f"\nfirst floor pixel at steps {first}")
    assert first["f32"] == first["f32alt"] == first["bf16"] and 5 <= first["f32"] < steps - 60, first
    t0 = first["f32"]
    scale = np.abs(f32[:t0]).max()
    pre_ctl, pre_bf = np.abs(falt - f32)[:t0].max(), np.abs(bf - f32)[:t0].max()
    assert pre_ctl <= 1e-3 * scale and pre_bf <= 2e-2 * scale, (pre_ctl, pre_bf, scale)
    post = slice(t0 + 10, None)
    m_ctl, m_bf = np.median(np.abs(falt - f32)[post]), np.median(np.abs(bf - f32)[post])
    print(f"before step {t0}: |f32alt-f32| <= {pre_ctl:.2e}, |bf16-f32| <= {pre_bf:.2e} (loss scale {scale:.3f}); after: "
          f"median |f32alt-f32| {m_ctl:.4f}, median |bf16-f32| {m_bf:.4f}; validation f32 {runs['f32'][2]:.4f} "
          f"f32alt {runs['f32alt'][2]:.4f} bf16 {runs['bf16'][2]:.4f}")
    assert m_ctl >= 20 * max(pre_ctl, 1e-6), (m_ctl, pre_ctl)
    assert m_bf <= 4 * m_ctl, (m_bf, m_ctl)


def test_bf16_gradients_are_bit_reproducible_at_full_batch():
    """No float atomics anywhere (slab sums, d(alpha) / d(bias) partials and their batched reductions run in fixed orders)
    and every LDS-DMA stage is waited for by count: the same 256-stamp gradient step on fresh engines must give
    bit-identical gradients and loss, run after run (a missed wait shows up here as a rare mismatch)."""
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch()
    B = 256
    x, y = synthetic_stamps(B, seed=31)
    ref = None
    for rep in range(6):
        eng = _engine(arch, B)
        eng.init(seed=9)
        eng.optimizer_reset(1e-4)
        eng.upload(0, x, y)
        out = eng.grad_step(0, first=0, B=B, seed=77)
        grads = {name: eng.get_grad(name) for name, _, tr in eng.specs if tr}
        eng.close()
        if ref is None:
            ref = (out, grads)
            continue
        assert out == ref[0], rep
        for name, g in grads.items():
            assert np.array_equal(g, ref[1][name]), (rep, name)


def test_bf16_decoder_bucket_on_the_comm_stream_changes_no_bit():
    """The bf16 backward finishes the decoder's reductions at the decoder / encoder boundary, all-reduces that bucket on
    the comm stream while the encoder is differentiated and (early Adam) updates it there; the encoder bucket follows at
    the end.  DV_FORCE_COMM=1 runs this with a 1-rank RCCL communicator on one GPU, DV_NO_EARLY_ADAM=1 and DV_NO_OVERLAP=1
    take the single-stream orders: all four must produce bit-identical parameters after three steps."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
x, y = synthetic_stamps(128, seed=3)
eng = E.Engine(E.make_config(max_batch=64, dtype=1))
eng.init(seed=4); eng.optimizer_reset(1e-4); eng.upload(0, x, y)
out = eng.train_steps(0, 0, 64, 3, seed=7)
names = ("dec/convt5/kernel", "dec/head/kernel", "dec/prelut3/alpha", "dec/convt0/bias", "dec/dense1/kernel",
         "enc/conv0/kernel", "enc/conv5/kernel", "enc/dense/kernel", "enc/bn/gamma")
print(repr(out["loss"]), " ".join(repr(float(np.abs(eng.get_param(n).astype(np.float64)).sum())) for n in names))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ({}, {"DV_FORCE_COMM": "1"}, {"DV_NO_EARLY_ADAM": "1"}, {"DV_NO_OVERLAP": "1"}):
        env = dict(os.environ)
        for k in ("DV_FORCE_COMM", "DV_NO_EARLY_ADAM", "DV_NO_OVERLAP"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2] == outs[3], outs


def test_bf16_queued_steps_at_full_batch_are_bit_reproducible_across_stream_orders():
    """Forty queued 256-stamp train steps (conv and dense-trunk weight gradients on the aux stream, their sums on the
    reduction stream, one launch on the main stream with its own slab region, early Adam per bucket on the comm stream)
    must leave bit-identical parameters on every run and in the single-stream order (DV_NO_OVERLAP=1).  Round 3: a slab
    region that the main stream rewrote before the aux stream had summed it showed up exactly here - a loss that changed
    from run to run at this batch size while the 64-stamp / 3-step test above stayed green."""
    import os
    import subprocess
    import sys
    import zlib

    code = r'''
import sys, zlib, numpy as np
sys.path.insert(0, %r)
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
x, y = synthetic_stamps(256, seed=0)
eng = E.Engine(E.make_config(max_batch=256, dtype=1))
eng.init(seed=4); eng.optimizer_reset(1e-4); eng.upload(0, x, y)
out = eng.train_steps(0, 0, 256, 40, seed=2)
crc = 0
for name, _, tr in eng.specs:
    if tr:
        crc = zlib.crc32(np.ascontiguousarray(eng.get_param(name)).tobytes(), crc)
print(repr(out["loss"]), crc)
eng.close()
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # ... and in the stream orders the product's A/B switches restore: the forms of the step's seams until round 6 (loss /
    # head-bias sums on the main stream, the tail of the pass on the weight-gradient stream), bucket-boundary sums on the
    # weight-gradient stream, the input normalised at the head of the step, a one-rank communicator
    orders = ({}, {}, {}, {"DV_NO_OVERLAP": "1"}, {"DV_BF_SUMS_ON_MAIN": "1"}, {"DV_BF_TAIL_ON_WGRAD_STREAM": "1"},
              {"DV_BF_SUMS_ON_MAIN": "1", "DV_BF_TAIL_ON_WGRAD_STREAM": "1", "DV_BF_BOUNDARY_ON_WGRAD_STREAM": "1",
               "DV_NO_INPUT_AHEAD": "1"}, {"DV_FORCE_COMM": "1"})
    switches = ("DV_FORCE_COMM", "DV_NO_EARLY_ADAM", "DV_NO_OVERLAP", "DV_BF_SUMS_ON_MAIN", "DV_BF_TAIL_ON_WGRAD_STREAM",
                "DV_BF_BOUNDARY_ON_WGRAD_STREAM", "DV_NO_INPUT_AHEAD")
    for extra in orders:
        env = dict(os.environ)
        for k in switches:
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, list(zip(orders, outs))


def test_python_surface_with_the_bf16_engine(tmp_path):
    """create_model_vae(..., dtype="bf16") behind the reference's call surface (train.py:27-37,118-130,
    deblender.py:6-24): compile / fit with validation / History, deblend on float64 stamps, TF-format checkpoint round trip
    (the checkpoint holds the fp32 master weights, so an fp32 net can load what a bf16 net saved)."""
    from debvader_amd.data import synthetic_stamps
    from debvader_amd.deblend_cutout.deblender import deblend
    from debvader_amd.model import model
    from debvader_amd.training.metrics import vae_loss

    arch = dict(input_shape=(59, 59, 6), latent_dim=32, filters=[32, 64, 128, 256], kernels=[3, 3, 3, 3])
    net, enc, dec, z = model.create_model_vae(**arch, max_batch=64, dtype="bf16", seed=2)
    net.compile(optimizer=model.Adam(learning_rate=1e-4), loss=vae_loss, metrics=["mse"])
    x, y = synthetic_stamps(96, seed=1)
    xv, yv = synthetic_stamps(20, seed=2)
    hist = net.fit(x, y, epochs=2, batch_size=64, verbose=0, validation_data=(xv, yv))
    assert sorted(hist.history) == ["loss", "mse", "val_loss", "val_mse"]
    assert all(np.isfinite(v).all() and len(v) == 2 for v in hist.history.values())
    mean, dist = deblend(net, x[:5].astype(np.float64))
    assert mean.shape == (5, 59, 59, 6) and np.isfinite(mean).all() and dist.stddev().numpy().min() >= 1e-4 * (1 - 1e-6)
    net.save_weights(str(tmp_path / "w" / "weights_noisy_v4.ckpt"))
    ref, _, _, _ = model.create_model_vae(**arch, max_batch=64, seed=3)          # fp32 engine
    ref.load_weights(model.latest_checkpoint(str(tmp_path / "w")))
    for a, b in zip(net.get_weights(), ref.get_weights()):
        np.testing.assert_array_equal(a, b)
    m32, _ = deblend(ref, x[:5])
    t_bf, t_32 = enc(x[:5]).numpy(), ref.encoder(x[:5]).numpy()
    assert np.abs(t_bf - t_32).max() <= 2e-2 * np.abs(t_32).max()              # the format's cost on the encoder output


def test_deep_arch_128px_six_levels():
    """BASELINE configs[3] on the bf16 engine: 128 x 128 x 6 stamps, six levels up to 512 channels (no crop: 128 = 2^7),
    two stamps against both oracles - the unfused forms of a 16-stamp block with two real stamps.  With TWO stamps a
    kernel gradient is a sum of two products and one flipped rounding in the 23 bf16 layers above it moves it by half its
    size (measured 0.57 * max on dec/dense1/kernel between the engine and the bf16 oracle, round 6), so this case only
    bounds gross errors; the bounds that mean something are the 64-stamp case below and the per-layer test."""
    arch = vo.Arch(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6)
    _run(arch, B=2, seed=5, tol_out=3e-2, tol_grad_b=0.75, min_cos=0.85, tol_grad_64=0.75)


def test_deep_arch_128px_at_its_per_gpu_batch_of_64():
    """the WHOLE bf16 step of the 128 x 128 x 6 / six-level net at BASELINE configs[3]'s per-GPU batch
    of 64 stamps (until round 5: two stamps) against the bf16-rounding oracle and the float64 oracle - outputs 1.5e-2 / 2e-2 *
    max, ELBO 5e-4 / 5e-3 relative, gradients per tensor within min(0.25, 2 x the bf16 oracle's own distance from float64)
    of the bf16 oracle and within 0.25 * max, cosine >= 0.97, of float64.  The oracle evaluations (two minutes on one core)
    come from a worker process (tests/oracle_pool.py)."""
    arch = vo.Arch(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6)
    _run(arch, B=64, seed=23, tol_out=1.5e-2, tol_grad_b=0.25, min_cos=0.97, tol_grad_64=0.25, per_tensor=True)


def test_thousand_stamp_batches_on_the_bf16_engine():
    """Per-GPU batches of 1024 stamps (16 slab rows per weight-gradient launch, more than the 24 MB slab budget holds for
    the 256-channel layers: the budget is a target, one slab per 64-stamp chunk the floor): the step runs, and its loss
    and gradients agree with the fp32 engine on the same batch as the 256-stamp case does."""
    from debvader_amd.data import synthetic_stamps

    arch = vo.Arch()
    B = 1024
    x, y = synthetic_stamps(B, seed=41)
    res = []
    for dtype in (0, 1):
        eng = _engine(arch, B, dtype=dtype)
        eng.init(seed=6)
        hb = eng.get_param("dec/head/bias")
        hb[arch.nb:] += 0.3
        eng.set_param("dec/head/bias", hb)
        eng.optimizer_reset(1e-4)
        eng.upload(0, x, y)
        out = eng.grad_step(0, first=0, B=B, seed=13)
        res.append((out, {n: eng.get_grad(n) for n in ("dec/convt0/kernel", "enc/conv7/kernel", "enc/conv0/kernel", "dec/head/bias")}))
        eng.close()
    (o32, g32), (obf, gbf) = res
    assert abs(obf["loss"] - o32["loss"]) <= 1e-2 * abs(o32["loss"])
    for n in g32:
        assert np.isfinite(gbf[n]).all()
        assert _cos(gbf[n], g32[n].astype(np.float64)) >= 0.97, (n, _cos(gbf[n], g32[n].astype(np.float64)))


def test_channel_counts_that_are_not_powers_of_two():
    """filters (32, 96) on the bf16 engine: 12 eight-channel pieces per row do not divide the 256 threads of the unfused
    PReLU-backward kernel (ragged batch), three 32-channel column tiles in the fused one (64 stamps)."""
    arch = vo.Arch(input_shape=(20, 20, 4), latent_dim=8, filters=(32, 96), kernels=(3, 3))
    # (the KL regulariser of this tiny net is ~1e-3: its bf16 cost is 0.55 % of itself, the stated 1 % here)
    _run(arch, B=5, seed=21, tol_grad_b=5e-2, tol_scal_64=1e-2)
    _run(arch, B=64, seed=22, tol_grad_b=0.1, min_cos=0.95, tol_grad_64=0.35, tol_scal_64=1e-2)


def test_two_row_weight_gradient_against_the_one_row_form_on_odd_shapes():
    """bwgrad2_kernel (stride-1 layers: two Y rows per wave, row pairs and - for wide rows - column segments per workgroup)
    against bwgrad_kernel (DV_BWGRAD_ONE_ROW=1, read once per process: two child processes) on shapes the BASELINE nets do not
    have: odd and small row counts (21, 11, 6 / 37, 19, 10, 5), a last pair with one row, 48 and 20 stamps (three and two
    16-stamp chunks, a workgroup with idle waves), 16-channel operands on both sides.  The two forms add the same bf16
    products in different orders into fp32: every kernel gradient agrees to 2e-5 of the tensor's largest entry, everything
    that is not a stride-1 kernel gradient bit for bit."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from debvader_amd import engine as E
from debvader_amd.data import synthetic_stamps
out = {}
for size, filters, B in ((21, (32, 64, 128), 48), (37, (32, 32, 64, 64), 20), (64, (32, 64), 16)):
    x, y = synthetic_stamps(B, seed=5, size=size, nb=6)
    eng = E.Engine(E.make_config((size, size, 6), 16, filters, (3,) * len(filters), max_batch=B, dtype=1))
    eng.init(seed=9); eng.upload(0, x, y)
    eng.grad_step(0, first=0, B=B, seed=3)
    for name, _, tr in eng.specs:
        if tr and name.endswith("/kernel"):
            out["%%d/%%s" %% (size, name)] = eng.get_grad(name)
    eng.close()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    res = []
    with tempfile.TemporaryDirectory() as td:
        for form in ("two", "one"):
            env = dict(os.environ)
            env.pop("DV_BWGRAD_ONE_ROW", None)
            if form == "one":
                env["DV_BWGRAD_ONE_ROW"] = "1"
            path = os.path.join(td, form + ".npz")
            r = subprocess.run([sys.executable, "-c", code % root, path], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            res.append(dict(np.load(path)))
    two, one = res
    assert sorted(two) == sorted(one) and len(two) >= 20
    differing = 0
    for k in two:
        scale = float(np.abs(one[k]).max())
        err = float(np.abs(two[k].astype(np.float64) - one[k]).max())
        assert scale > 0 and err <= 2e-5 * scale, (k, err, scale)
        differing += int(err > 0)
    assert differing >= 6          # (the stride-1 layers really took the other kernel)


def test_head_in_the_head_convs_epilogue_against_the_head_kernel():
    """A bf16 train / gradient step that keeps no outputs runs the head (relu, crop, sigma floor, Normal NLL and its gradient;
    model.py:139-161, train.py:27-37) in the epilogue of the head conv (BEPI_HEAD); with kept outputs - or DV_BF_HEAD_FUSED=0 -
    the head conv writes its fp32 tensor and bf_head_kernel reads it.  Same arithmetic per element: every parameter gradient
    must be bit-identical between the two, the loss sums (other partial sums) to 1e-6; also with the Keras sample-MSE metric
    (Philox draws per element) and on a batch with pad stamps (200 of 256)."""
    from debvader_amd import engine as E
    from debvader_amd.data import synthetic_stamps

    for B, sample in ((256, False), (200, True), (64, False)):
        x, y = synthetic_stamps(B, seed=21)
        eng = E.Engine(E.make_config(max_batch=B, dtype=1))
        eng.init(seed=2)
        hb = eng.get_param("dec/head/bias")                 # (zero at initialisation: a head without its bias would pass)
        hb += np.linspace(-0.2, 0.4, hb.size).astype(np.float32)
        eng.set_param("dec/head/bias", hb)
        eng.upload(0, x, y)
        eng.set_mse_sample(sample)
        eng.keep_outputs(True)
        ref = eng.grad_step(0, first=0, B=B, seed=13)
        gref = {n: eng.get_grad(n).copy() for n, _, tr in eng.specs if tr}
        eng.keep_outputs(False)
        out = eng.grad_step(0, first=0, B=B, seed=13)
        for k in ("loss", "nll_mean", "mse"):
            assert abs(out[k] - ref[k]) <= 1e-6 * abs(ref[k]), (B, k, out[k], ref[k])
        for n, g in gref.items():
            assert np.array_equal(eng.get_grad(n), g), (B, n)
        eng.close()
