"""GPU parity: HIP engine (through the C-ABI) vs the fp64 oracle on identical inputs, weights and eps.

Tolerances (stated here, checked below):
  * activations / outputs (t, z, kl, loc, scale): |gpu - oracle| <= 2e-4 * max|oracle| per tensor
  * ELBO scalars (loss, nll_mean, kl_reg): relative error <= 1e-4   (BASELINE.json north_star)
  * gradients: |gpu - oracle| <= 1e-3 * max|oracle| per tensor (2e-3 for the 2-6 element d(gamma) / d(beta) of the input
    BatchNorm: see _grad_tol for the measurements behind it)
  * parameters after one legacy-Adam step: |gpu - oracle| <= 2e-6 absolute (lr = 1e-4)
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from tests import margins

pytestmark = pytest.mark.gpu


def _engine(arch, max_batch):
    from debvader_amd import engine as E

    cfg = E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels),
                        max_batch=max_batch)
    return E.Engine(cfg)


def small_arch():
    # odd sizes 13 -> 7 -> 4, both SAME-pad cases, odd crop (16 - 13 = 3 -> (1,2))
    return vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(8, 16), kernels=(3, 3))


def _relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _case(arch, B, seed, data=None, sigma_bias=0.0):
    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    p["dec/head/bias"][arch.nb:] += sigma_bias      # > 0: sigma off its 1e-4 floor (tests/test_gpu_0_fullsize_oracle.py)
    H, W, C = arch.input_shape
    if data is None:
        x = rng.normal(0, 0.4, size=(B, H, W, C)).astype(np.float32)
        y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C))).astype(np.float32)
    else:
        x, y = data
    eps = rng.normal(size=(B, arch.latent_dim)).astype(np.float32)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return p, x, y, eps


def _grad_tol(name):
    """Gradient tolerance relative to the tensor's largest element: 1e-3, and 2e-3 for the two input-BatchNorm tensors.
    d(gamma) / d(beta) are 2-6 numbers, each a sum over every first-layer weight gradient with cancellation of ~1e4: on the
    3-stamp toy case of test_channel_counts_that_are_not_powers_of_two ANY float32 evaluation scatters between 2e-4 and
    1.1e-3 around the float64 oracle (numpy float32 oracle 3.1e-4, torch CPU float32 autograd 7.9e-4, this engine's direct
    kernels 2.3e-4, its Winograd kernels 5.6e-4 / 1.09e-3), so a 1e-3 bar sits inside float32's own noise there.  Every
    other tensor - and these two on the real 59-px architecture, measured <= 4.2e-4 - keeps 1e-3."""
    return 2e-3 if name in ("enc/bn/gamma", "enc/bn/beta") else 1e-3


def _gate_names(arch):
    """every tensor whose sign gates a derivative: the pre-activations of all PReLUs and of the head's relu (oracle names)"""
    n2 = 2 * len(arch.filters)
    return [f"enc_u{j}" for j in range(n2)] + ["enc_flat_u", "dec_z", "dec_u_h", "dec_u_r"] + [f"dec_u{j}" for j in range(n2)] + ["head_pre"]


def _engine_gate_tensor(eng, arch, B, name):
    """the engine's own copy of that tensor (float32), in the oracle's shape"""
    L2 = 2 * len(arch.filters)
    fl = arch.filters[-1]
    H, W, C = arch.input_shape
    if name.startswith("enc_u") and name[5:].isdigit():
        j = int(name[5:])
        lvl = j // 2
        h = arch.enc_sizes[lvl + 1] if j % 2 else arch.enc_sizes[lvl]
        return eng.activation(name, (B, h, h, arch.filters[lvl]))
    if name.startswith("dec_u") and name[5:].isdigit():
        j = int(name[5:])
        lvl = len(arch.filters) - 1 - j // 2
        h = arch.w0 * 2 ** (j // 2 + 1)
        return eng.activation(name, (B, h, h, arch.filters[lvl]))
    if name == "enc_flat_u":
        s = arch.enc_sizes[-1]
        return eng.activation(f"enc_a{L2 - 1}", (B, s, s, fl)).reshape(B, -1)
    if name == "dec_z":
        return eng.activation("z", (B, arch.latent_dim))
    if name == "dec_u_h":
        return eng.activation("dec_uh", (B, arch.dec_hidden))
    if name == "dec_u_r":
        return eng.activation("dec_ur", (B, arch.w0 * arch.w0 * fl))
    if name == "head_pre":
        return eng.activation("head_pre", (B, arch.dec_out, arch.dec_out, 2 * C))
    raise KeyError(name)


def _gate_matched_gradients(eng, arch, p, x, y, eps, B, train_decoder):
    """The float64 oracle evaluated AT THE ENGINE'S GATE STATES: float64 forward and backward, but every PReLU / relu gate
    (u > 0) whose state differs from the engine's takes the engine's - nothing else of the engine's arithmetic enters.
    A gate is a discontinuity of the derivative: where a pre-activation lies within float32 rounding of zero (a few
    hundred of 2.5e8 on the 128-px net: profiles/r06_gate_flip_probe.txt) two correct evaluations may sit on different
    sides, and the gradient they compute differs by that unit's whole contribution - up to 4e-2 * max on the early encoder
    tensors, although every sum is computed to float32 rounding.  With the gates matched what is left is rounding.
    Returns (gradients, number of gates that differed)."""
    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
    c = vo.forward(arch, p, x64, e64, training=True)
    flips = 0
    for n in _gate_names(arch):
        ge = _engine_gate_tensor(eng, arch, B, n) > 0
        a = c[n]
        m = (a > 0) != ge.reshape(a.shape)
        k = int(m.sum())
        if k:
            a = a.copy()
            a[m] = np.where(ge.reshape(a.shape)[m], 1e-300, -1e-300)
            c[n] = a
            flips += k
    return vo.backward(arch, p, c, y64, train_decoder=train_decoder), flips


def _run_parity(arch, B, seed, data=None, train_decoder=True, sigma_bias=0.0, f32_floor=False, data_seed=None,
                gate_matched=False):
    """f32_floor: gradient tolerance per tensor = max(_grad_tol, min(1.5 x the error of a numpy float32 evaluation of the
    same step against the float64 oracle, 1e-2)) - at the quoted batch sizes a gradient is a sum over ~10^6 signed pixel
    terms that went through 25 (59 px) or 37 (128 px) layers, and a plain float32 evaluation misses float64 by up to
    2e-2 * max on some tensors (measured on MI355X boxes / here: enc/conv4/kernel 1.8e-2, enc/prelu1/alpha 1.7e-2 at 256
    stamps where the engine is at 1.5e-3; dec/prelu_in/alpha of the 128-px net at 64 stamps 3.64e-3 where the engine is at
    3.63e-3 - the same conditioning, two implementations).  So: within 1e-3, or no worse than an independent float32
    evaluation of the same formulas (1.5 x: two float32 orders scatter around each other), and never beyond 1e-2.  The
    tight per-layer bound (<= 2e-5 of every tensor, nothing cascades) is tests/test_gpu_0_layers_f32.py's."""
    from tests import oracle_jobs, oracle_pool

    if data_seed is not None:
        data = oracle_jobs.stamps(B, data_seed)
    p, x, y, eps = _case(arch, B, seed, data, sigma_bias)
    eng = _engine(arch, max_batch=B)
    eng.set_params(p)
    eng.set_trainable(True, train_decoder)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)

    # the oracle: float64 forward / losses / backward (and the numpy-float32 evaluation of the same step for f32_floor).
    # Cases built from seeds alone may have been evaluated ahead by a worker process (tests/oracle_pool.py)
    if (data is None or data_seed is not None) and arch == oracle_jobs.make_arch(oracle_jobs.arch_kw(arch)):
        ev = oracle_pool.fetch("f32_case", arch_kw=oracle_jobs.arch_kw(arch), B=B, seed=seed, data_seed=data_seed,
                               sigma_bias=sigma_bias, train_decoder=train_decoder, f32_floor=f32_floor)
    else:
        ev = oracle_jobs.f32_eval(arch, p, x, y, eps, train_decoder, f32_floor)
    c, ref, g, floor = dict(ev["acts"], **ev["bn"]), ev["ref"], ev["g"], ev["floor"]

    def tol(name):
        return max(_grad_tol(name), min(1.5 * floor.get(name, 0.0), 1e-2))

    eng.keep_outputs(True)          # loc / scale of the step are compared below
    out = eng.grad_step(0, first=0, B=B, eps=eps)
    d = arch.latent_dim
    H, W, C = arch.input_shape
    acts = {
        "t": eng.activation("t", (B, arch.params_size)),
        "z": eng.activation("z", (B, d)),
        "kl": eng.activation("kl", (B,)),
        "loc": eng.activation("loc", (B, H, W, C)),
        "scale": eng.activation("scale", (B, H, W, C)),
    }
    for k, v in acts.items():
        assert _relmax(v, c[k]) <= 2e-4, (k, _relmax(v, c[k]))
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        assert abs(out[k] - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-12, (k, out[k], ref[k])
    worst = ("", 0.0)
    rows = [(k, _relmax(acts[k], c[k]), None, 2e-4, "activation / output") for k in acts]
    rows += [(k, abs(out[k] - ref[k]) / (abs(ref[k]) + 1e-30), None, 1e-4, "ELBO scalar, relative") for k in ("loss", "nll_mean", "kl_reg")]
    failed = []
    for name, _, tr in arch.param_specs():
        if name not in g:
            continue
        e = _relmax(eng.get_grad(name), g[name])
        if e > worst[1]:
            worst = (name, e)
        rows.append((name, e, floor.get(name), tol(name),
                     "gradient" + ("" if tol(name) <= _grad_tol(name) else "  [bound above 1e-3: float32's own distance]")))
        if e > tol(name):
            failed.append((name, e, tol(name)))
    margins.record(f"fp32 engine vs float64 oracle: {'x'.join(map(str, arch.input_shape))}, {len(arch.filters)} levels, B={B}, "
                   f"{'stage 1' if train_decoder else 'stage 2 (decoder frozen)'}, head scale bias +{sigma_bias}", rows,
                   "errors are max|a - oracle| / max|oracle| per tensor; 'other impl' = the numpy float32 evaluation of the same step")
    assert not failed, failed
    if gate_matched:
        # ... and the bound WITHOUT the float32-floor exception: against the float64 oracle evaluated at the engine's own
        # gate states every gradient tensor holds the file header's 1e-3 (2e-3 for the two BatchNorm tensors)
        gm, nflip = _gate_matched_gradients(eng, arch, p, x, y, eps, B, train_decoder)
        rows_g, bad = [], []
        for name in g:
            e = _relmax(eng.get_grad(name), gm[name])
            rows_g.append((name, e, _relmax(g[name], gm[name]), _grad_tol(name), "gradient vs the gate-matched float64 oracle"))
            if e > _grad_tol(name):
                bad.append((name, e))
        margins.record(f"fp32 engine vs float64 oracle AT THE ENGINE'S GATE STATES ({nflip} gates differ from float64's): "
                       f"{'x'.join(map(str, arch.input_shape))}, {len(arch.filters)} levels, B={B}, head scale bias +{sigma_bias}",
                       rows_g, "'other impl' = what the differing gates alone are worth (plain float64 vs gate-matched float64)")
        print(f"\n  {nflip} gates differ between the engine and float64; gate-matched: largest gradient error "
              f"{max(r[1] for r in rows_g):.2e} * max")
        assert not bad, bad

    # the production form of the step: no loc / scale stores in the head kernel
    eng.keep_outputs(False)
    outf = eng.grad_step(0, first=0, B=B, eps=eps)
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        assert abs(outf[k] - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-12, (k, outf[k], ref[k])
    for name in g:
        e = _relmax(eng.get_grad(name), g[name])
        assert e <= tol(name), ("without outputs", name, e)
    eng.keep_outputs(True)
    eng.grad_step(0, first=0, B=B, eps=eps)          # gradients of the form the train step below repeats

    # one full training step: Adam update + BN moving statistics.  The first Adam step moves every weight by
    # ~lr*sign(g), so weights whose gradient is at rounding level may legitimately differ by 2*lr from the
    # oracle's update; the update itself is therefore checked against the oracle's Adam fed with the
    # engine's own gradients (the step recomputes them bit-identically), and against the oracle's
    # gradients only where those are resolved (|g| > 1e-2 max|g|).
    gg = {name: eng.get_grad(name).astype(np.float64) for name in g}
    st = vo.AdamState()
    p2 = {k: v.copy() for k, v in p.items()}
    vo.adam_step(st, p2, gg)
    st3 = vo.AdamState()
    p3 = {k: v.copy() for k, v in p.items()}
    vo.adam_step(st3, p3, g)
    vo.bn_moving_update(arch, p2, c)
    out2 = eng.train_step(0, first=0, B=B, eps=eps)
    assert abs(out2["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"])
    for name, _, tr in arch.param_specs():
        if tr and name not in g:
            np.testing.assert_array_equal(eng.get_param(name), p[name].astype(np.float32))   # frozen decoder
            continue
        got = eng.get_param(name)
        assert np.abs(got - p2[name]).max() <= 1e-6, (name, np.abs(got - p2[name]).max())
        if name in g:
            res = np.abs(g[name]) > 1e-2 * np.abs(g[name]).max()
            assert np.abs(got - p3[name])[res].max() <= 1e-6, name
            np.testing.assert_allclose(eng.get_slot(name, 0), st.m[name], rtol=5e-5, atol=1e-30)  # fp32 (1-beta) as in TF
            np.testing.assert_allclose(eng.get_slot(name, 1), st.v[name], rtol=5e-5, atol=1e-30)
    assert eng.iterations == 1
    eng.close()
    if floor:
        wf = max(floor.items(), key=lambda kv: kv[1])
        print(f"\n  numpy float32 evaluation of the same step: largest gradient error {wf[1]:.2e} * max ({wf[0]}); "
              f"engine: {worst[1]:.2e} ({worst[0]}; numpy float32 there: {floor[worst[0]]:.2e})")
    return worst


def test_keras_mse_metric_is_taken_against_a_sample_of_the_output_distribution():
    """compile(metrics=["mse"]) of the reference (train.py:128) measures the labels against a SAMPLE of Normal(loc, scale)
    (model.py:158).  With dv_model_set_mse_sample the engine draws that sample from its Philox stream; the oracle
    reproduces the draw (vo.philox_normal) - for the fp32 and the bf16 engine, several forward lanes included."""
    from debvader_amd import engine as E

    for dtype, arch, B in ((0, small_arch(), 7), (0, small_arch(), 96),
                           (1, vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3)), 21)):
        p, x, y, eps = _case(arch, B, 12)
        eng = E.Engine(E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels),
                                     max_batch=B, dtype=dtype))
        eng.set_params(p)
        eng.optimizer_reset(1e-4)
        eng.upload(0, x, y)
        eng.keep_outputs(True)
        seed = 987654321012
        plain = eng.grad_step(0, first=0, B=B, eps=eps, seed=seed)
        eng.set_mse_sample(True)
        out = eng.grad_step(0, first=0, B=B, eps=eps, seed=seed)
        H, W, C = arch.input_shape
        loc = eng.activation("loc", (B, H, W, C)).astype(np.float64)
        scale = eng.activation("scale", (B, H, W, C)).astype(np.float64)
        e = vo.philox_normal(seed, 0x4D534500, B, H * W * C).astype(np.float64).reshape(B, H, W, C)
        ref = ((y - (loc + scale * e)) ** 2).mean()
        assert abs(out["mse"] - ref) <= 1e-4 * ref, (dtype, B, out["mse"], ref)
        assert abs(plain["mse"] - ((y - loc) ** 2).mean()) <= 1e-4 * plain["mse"]
        assert out["loss"] == plain["loss"]                     # the metric does not touch the loss
        eng.close()


def test_small_arch_parity():
    _run_parity(small_arch(), B=5, seed=0)


def test_small_arch_parity_ragged_batch_and_frozen_decoder():
    # B=3 is not a multiple of any tile; stage 2 of train_deblender freezes the decoder (train.py:175)
    _run_parity(small_arch(), B=3, seed=4, train_decoder=False)


def test_full_arch_parity_b4():
    from debvader_amd.data import synthetic_stamps

    x, y = synthetic_stamps(4, seed=5)
    _run_parity(vo.Arch(), B=4, seed=2, data=(x, y), gate_matched=True)


def test_deeper_128px_arch_parity():
    # BASELINE configs[3] shape: 128x128x6 stamps, 6 levels (filters as in SURVEY 8(d)): exercises 512-channel
    # layers, no crop (128 = 2^7), wide rows (strip kernels must fit or fall back)
    arch = vo.Arch(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6)
    assert arch.crop == (0, 0) and arch.flat == 2 * 2 * 512
    _run_parity(arch, B=2, seed=21)


def test_inference_matches_oracle_and_is_stochastic():
    arch = small_arch()
    p, x, y, eps = _case(arch, 6, seed=9)
    eng = _engine(arch, max_batch=4)       # forces two chunks (4 + 2)
    eng.set_params(p)
    r = eng.infer(x, eps=eps, want=("loc", "scale", "mu", "zstd", "z"))
    c = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=False)
    assert _relmax(r["loc"], c["loc"]) <= 2e-4
    assert _relmax(r["scale"], c["scale"]) <= 2e-4
    assert _relmax(r["mu"], c["mu"]) <= 2e-4
    assert _relmax(r["z"], c["z"]) <= 2e-4
    std = np.sqrt((c["L"] ** 2).sum(-1))
    assert _relmax(r["zstd"], std) <= 2e-4
    # encoder / decoder sub-models
    assert _relmax(eng.encode(x), c["t"]) <= 2e-4
    loc, scale = eng.decode(c["z"].astype(np.float32))
    assert _relmax(loc, c["loc"]) <= 2e-4 and _relmax(scale, c["scale"]) <= 2e-4
    # engine-drawn eps: different seeds give different samples, same seed reproduces bit for bit
    a = eng.infer(x, seed=1, want=("loc", "z"))
    b = eng.infer(x, seed=2, want=("loc", "z"))
    a2 = eng.infer(x, seed=1, want=("loc", "z"))
    assert np.abs(a["z"] - b["z"]).max() > 1e-3
    np.testing.assert_array_equal(a["loc"], a2["loc"])
    eng.close()


def test_engine_eps_matches_philox_restatement():
    arch = small_arch()
    p, x, y, _ = _case(arch, 7, seed=11)
    eng = _engine(arch, max_batch=8)
    eng.set_params(p)
    r = eng.infer(x, seed=1234, want=("z", "mu"))
    eps_dev = eng.activation("eps", (7, arch.latent_dim))
    ref = vo.philox_normal(1234, 0, 7, arch.latent_dim)
    np.testing.assert_allclose(eps_dev, ref, rtol=0, atol=2e-5)
    eng.close()


def test_eval_step_uses_moving_statistics():
    arch = small_arch()
    p, x, y, eps = _case(arch, 4, seed=13)
    eng = _engine(arch, max_batch=4)
    eng.set_params(p)
    eng.upload(1, x, y)
    out = eng.eval_step(1, first=0, B=4, eps=eps)
    c = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=False)
    ref = vo.losses(arch, c, y.astype(np.float64))
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        assert abs(out[k] - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-12, (k, out[k], ref[k])
    eng.close()


def test_index_gather_equals_contiguous():
    arch = small_arch()
    p, x, y, eps = _case(arch, 6, seed=17)
    eng = _engine(arch, max_batch=3)
    eng.set_params(p)
    eng.upload(0, x, y)
    idx = np.array([4, 0, 5], dtype=np.int32)
    a = eng.grad_step(0, idx=idx, eps=eps[:3])
    ga = eng.get_grad("enc/conv1/kernel")
    eng.upload(0, x[idx], y[idx])
    b = eng.grad_step(0, first=0, B=3, eps=eps[:3])
    gb = eng.get_grad("enc/conv1/kernel")
    assert a == b
    np.testing.assert_array_equal(ga, gb)
    eng.close()


def test_bad_arguments_fail_loudly():
    from debvader_amd._lib import DvError

    arch = small_arch()
    eng = _engine(arch, max_batch=2)
    with pytest.raises(DvError):
        eng.train_step(0, first=0, B=2)            # no data uploaded
    x = np.zeros((2, 13, 13, 4), np.float32)
    eng.upload(0, x, x)
    with pytest.raises(DvError):
        eng.train_step(0, first=1, B=2)            # rows out of range
    with pytest.raises(DvError):
        eng.train_step(0, first=0, B=3)            # above max_batch
    with pytest.raises(ValueError):
        eng.infer(np.zeros((1, 12, 13, 4), np.float32))
    eng.close()


def test_specialised_kernels_match_the_general_gather_gemm():
    """gconv_strip / gconv_strip8 (stride-1, <= 32 channels, W <= 64) and gconv_s2 (fused stride-2 classes) against gconv2 on the
    same random operands, including image sizes that are not multiples of the strip / tile geometry, both weight
    layouts and every epilogue.  fp32 sums in a different order: 2e-5 of the largest output."""
    import ctypes as C
    from debvader_amd import engine as E
    from debvader_amd._lib import check
    from tests import debug_lib
    out = (C.c_float * 2)()
    cases = []
    for H in (64, 59, 40, 17, 8):
        for (cs, ct) in ((32, 32), (32, 16), (16, 32)):
            for dgrad, nmajor in ((0, 0), (1, 1)):
                if cs == 16 and not dgrad:
                    continue
                for epi in (0, 1, 2):
                    cases.append((3, H, cs, H, ct, 1, 1, dgrad, nmajor, epi))
    # first-layer strip form (8 physical input channels, k-major weights, forward only)
    for H in (64, 59, 33, 9):
        for epi in (0, 1, 2):
            cases.append((3, H, 8, H, 32, 1, 1, 0, 0, epi))
    # fused stride-2 form: Conv2DTranspose forward (out = 2 in, pad 0) and Conv2D data gradient (odd sizes, pad 0 / 1)
    for (hs, cs, ht, ct, pb) in ((8, 64, 16, 32, 0), (16, 128, 32, 64, 0), (30, 32, 59, 32, 1), (15, 64, 30, 64, 0),
                                 (4, 256, 8, 256, 0)):
        for epi in (0, 2):
            cases.append((3, hs, cs, ht, ct, 2, pb, 1, 1, epi))
    # stride-1 layers with >= 32 channels on both sides take the Winograd kernel by default (next test): switch it off so
    # that the strip forms are what is checked here
    with debug_lib.debug_build() as lib:        # the cross-check harness lives in libdebvader_hip_debug.so
        ctx = E.default_context()
        check(lib.dv_debug_winograd(0))
        try:
            for c in cases:
                check(lib.dv_debug_gconv_check(ctx._h, *c, out))
                assert out[1] > 0.1, c
                assert out[0] <= 2e-5 * out[1], (c, out[0], out[1])
        finally:
            check(lib.dv_debug_winograd(1))


def test_winograd_kernel_matches_the_general_gather_gemm():
    """wino.hip (F(2x2, 3x3) for the stride-1 layers with >= 32 channels: model.py:81-83,128-134 and their data gradients)
    against gconv2 on the same random operands: every image size of the two architectures' stride-1 layers plus sizes
    that leave partial 8 x 8 blocks and partial tiles, channel counts 32 ... 256 with one to eight 32-column tiles and two
    to sixteen K chunks, both weight layouts / tap orders, every epilogue.  Same fp32 products in another association
    (sums of four inputs, halves in G): stated 2e-5 of the largest output, measured <= 2.7e-6."""
    import ctypes as C
    from debvader_amd import engine as E
    from debvader_amd._lib import check
    from tests import debug_lib
    out = (C.c_float * 2)()
    with debug_lib.debug_build() as lib:
        worst = _winograd_cases(lib, E.default_context(), check, out)
    print(f"\nWinograd vs gather-GEMM: worst relative difference {worst:.2e}")


def _winograd_cases(lib, ctx, check, out):
    worst = 0.0
    for H in (64, 59, 40, 32, 30, 17, 16, 15, 8, 5):
        for (cs, ct) in ((32, 32), (64, 64), (32, 64), (64, 32), (64, 128), (128, 128), (128, 256), (256, 256), (96, 160)):
            if H > 32 and cs * ct > 64 * 64:
                continue
            for dgrad, nmajor in ((0, 0), (1, 1)):
                for epi in (0, 1, 2):
                    c = (3, H, cs, H, ct, 1, 1, dgrad, nmajor, epi)
                    check(lib.dv_debug_gconv_check(ctx._h, *c, out))
                    assert out[1] > 0.1, c
                    assert out[0] <= 2e-5 * out[1], (c, out[0], out[1])
                    worst = max(worst, out[0] / out[1])
    # several items per workgroup (the LDS ring and, in the four-wave kernel, the two DMA streams run across item
    # boundaries; >= 48 input channels take the four-wave kernel, 32 the eight-wave one), ragged block groups, Cout = 48
    for (NB, H, cs, ct) in ((37, 30, 32, 64), (21, 64, 32, 32), (150, 8, 128, 256), (150, 8, 256, 256), (61, 16, 128, 128),
                            (45, 15, 64, 128), (29, 59, 32, 48), (40, 10, 96, 96), (5, 13, 48, 96), (90, 17, 48, 32)):
        for dgrad, nmajor in ((0, 0), (1, 1)):
            for epi in (0, 2):
                c = (NB, H, cs, H, ct, 1, 1, dgrad, nmajor, epi)
                check(lib.dv_debug_gconv_check(ctx._h, *c, out))
                assert out[1] > 0.1, c
                assert out[0] <= 2e-5 * out[1], (c, out[0], out[1])
                worst = max(worst, out[0] / out[1])
    return worst


def test_winograd_weight_gradient_matches_the_direct_kernels():
    """wino_wgrad_kernel (F(3x3, 2x2): weight gradients of the stride-1 layers with >= 64 channels on both sides,
    model.py:81-83,128-134) against the direct weight-gradient kernels on the same random operands: the image sizes of
    those layers and sizes with partial blocks / tiles, one to sixteen 64 x 64 output tiles, batch sizes that give the
    splits uneven block ranges.  fp32 sums in another association: stated 2e-5 of the largest gradient, measured 7e-7."""
    import ctypes as C
    from debvader_amd import engine as E
    from debvader_amd._lib import check
    from tests import debug_lib
    out = (C.c_float * 2)()
    with debug_lib.debug_build() as lib:
        worst = _winograd_wgrad_cases(lib, E.default_context(), check, out)
    print(f"\nWinograd weight gradient vs direct: worst relative difference {worst:.2e}")


def _winograd_wgrad_cases(lib, ctx, check, out):
    worst = 0.0
    for H in (32, 30, 17, 16, 15, 8, 5):
        for (cx, cy) in ((64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (256, 256), (192, 64)):
            if H > 16 and cx * cy > 128 * 128:
                continue
            for NB in (3, 7, 40):
                if NB == 40 and H > 16:
                    continue
                check(lib.dv_debug_wgrad_check(ctx._h, NB, H, cx, cy, out))
                assert out[1] > 0.1, (H, cx, cy, NB)
                assert out[0] <= 2e-5 * out[1], (H, cx, cy, NB, out[0], out[1])
                worst = max(worst, out[0] / out[1])
    return worst


def test_channel_counts_that_are_not_powers_of_two():
    """filters (32, 96): 96 divides neither 1024 (the block size of the PReLU-backward bias sums) nor a column-tile width;
    such architectures used to be accepted and then fail in the first training step (found by tools/fuzz_configs.py).
    Full parity at the usual fp32 tolerances, 3 and 40 stamps."""
    arch = vo.Arch(input_shape=(20, 20, 4), latent_dim=8, filters=(32, 96), kernels=(3, 3))
    _run_parity(arch, B=3, seed=21)
    _run_parity(arch, B=40, seed=22)


def test_wide_latent_space_and_odd_filter_counts():
    """latent_dim 64 (params_size 2144 columns: more than one 1024-column tile of the bias-gradient column sums) and
    filters (24, 48) on 27-pixel stamps with 2 bands - configurations the sweep tool found failing or never exercised."""
    _run_parity(vo.Arch(input_shape=(20, 20, 2), latent_dim=64, filters=(32, 64), kernels=(3, 3)), B=5, seed=31)
    _run_parity(vo.Arch(input_shape=(27, 27, 4), latent_dim=8, filters=(24, 48), kernels=(3, 3)), B=6, seed=32)
