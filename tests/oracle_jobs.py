"""CPU-oracle evaluations of the whole-step GPU tests as plain functions of a few scalars (TEST INFRASTRUCTURE).

Each function rebuilds its case from seeds (parameters, stamps, noise: the same code the test itself runs), evaluates the
float64 oracle - and the numpy-float32 or the bf16-rounding evaluation the test compares with - and returns only what
the test looks at.  tests/oracle_pool.py runs the large ones (256 stamps of the 59-px net, 64 stamps of the 128-px net:
40 - 70 s each) in worker processes beside the GPU tests; called directly they are what the tests ran inline until round 5.
Reference semantics: oracle/vae_oracle.py (model.py:43-58,61-161, metrics.py:16-26, train.py:27-37)."""
import numpy as np

ACT_KEYS = ("t", "z", "kl", "loc", "scale")


def make_arch(arch_kw):
    from oracle import vae_oracle as vo

    return vo.Arch(**{k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in arch_kw.items()})


def arch_kw(arch):
    return {"input_shape": tuple(arch.input_shape), "latent_dim": arch.latent_dim, "filters": tuple(arch.filters),
            "kernels": tuple(arch.kernels)}


def stamps(B, data_seed):
    from debvader_amd.data import synthetic_stamps

    return synthetic_stamps(B, seed=data_seed)


def f32_case_inputs(arch, B, seed, data_seed=None, sigma_bias=0.0):
    """tests/test_gpu_parity.py::_case"""
    from oracle import vae_oracle as vo

    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    p["dec/head/bias"][arch.nb:] += sigma_bias      # > 0: sigma off its 1e-4 floor (tests/test_gpu_0_fullsize_oracle.py)
    H, W, C = arch.input_shape
    if data_seed is None:
        x = rng.normal(0, 0.4, size=(B, H, W, C)).astype(np.float32)
        y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C))).astype(np.float32)
    else:
        x, y = stamps(B, data_seed)
    eps = rng.normal(size=(B, arch.latent_dim)).astype(np.float32)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return p, x, y, eps


def _relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def f32_eval(arch, p, x, y, eps, train_decoder=True, f32_floor=False):
    """float64 oracle of one gradient step (+ the numpy-float32 evaluation's distance from it, per gradient tensor:
    what float32 itself costs on this case - float32 parameters, activations, BLAS accumulation; an independent fp32
    evaluation, not the engine)."""
    from oracle import vae_oracle as vo

    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
    c = vo.forward(arch, p, x64, e64, training=True)
    ref = vo.losses(arch, c, y64)
    g = vo.backward(arch, p, c, y64, train_decoder=train_decoder)
    floor = {}
    if f32_floor:
        p32 = {k: v.astype(np.float32) for k, v in p.items()}
        c32 = vo.forward(arch, p32, x.astype(np.float32), eps.astype(np.float32), training=True)
        g32 = vo.backward(arch, p32, c32, y.astype(np.float32), train_decoder=train_decoder)
        floor = {k: _relmax(g32[k], g[k]) for k in g}
    return {"acts": {k: c[k] for k in ACT_KEYS}, "ref": ref, "g": g, "floor": floor,
            "bn": {"bn_mean": c["bn_mean"], "bn_var": c["bn_var"], "x": np.empty(tuple(c["x"].shape[:3]) + (0,))}}


def f32_case(arch_kw, B, seed, data_seed=None, sigma_bias=0.0, train_decoder=True, f32_floor=False):
    arch = make_arch(arch_kw)
    p, x, y, eps = f32_case_inputs(arch, B, seed, data_seed, sigma_bias)
    return f32_eval(arch, p, x, y, eps, train_decoder, f32_floor)


def bf16_case_inputs(arch, B, seed, data_seed=None):
    """tests/test_gpu_bf16.py::_case"""
    from oracle import vae_oracle as vo

    rng = np.random.default_rng(seed)
    p = vo.init_params(arch, seed=seed + 1, perturb=0.05)
    H, W, C = arch.input_shape
    if data_seed is None:
        x = rng.normal(0, 0.4, size=(B, H, W, C)).astype(np.float32)
        y = np.abs(rng.normal(0, 0.4, size=(B, H, W, C))).astype(np.float32)
    else:
        x, y = stamps(B, data_seed)
    eps = rng.normal(size=(B, arch.latent_dim)).astype(np.float32)
    p["dec/head/bias"][arch.nb:] += 0.3
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    return p, x, y, eps


BF16_ACT_KEYS = ("t", "z", "kl", "loc", "scale", "head_pre")


def bf16_eval(arch, p, x, y, eps, train_decoder=True):
    """bf16-rounding oracle and float64 oracle of one gradient step of the bf16 engine's test case."""
    from oracle import vae_oracle as vo
    from oracle import vae_oracle_bf16 as vb

    B = x.shape[0]
    x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
    fused = ((B + 15) // 16 * 16) % 64 == 0
    cb = vb.forward(arch, p, x64, e64, training=True)
    rb = vo.losses(arch, cb, y64)
    gb = vb.backward(arch, p, cb, y64, train_decoder=train_decoder, fused=fused)
    c = vo.forward(arch, p, x64, e64, training=True)
    r = vo.losses(arch, c, y64)
    g = vo.backward(arch, p, c, y64, train_decoder=train_decoder)
    return {"cb": {k: cb[k] for k in BF16_ACT_KEYS}, "rb": rb, "gb": gb, "c": {k: c[k] for k in BF16_ACT_KEYS}, "r": r, "g": g}


def bf16_case(arch_kw, B, seed, data_seed=None, train_decoder=True):
    arch = make_arch(arch_kw)
    p, x, y, eps = bf16_case_inputs(arch, B, seed, data_seed)
    return bf16_eval(arch, p, x, y, eps, train_decoder)


FULL = {"input_shape": (59, 59, 6), "latent_dim": 32, "filters": (32, 64, 128, 256), "kernels": (3, 3, 3, 3)}
DEEP = {"input_shape": (128, 128, 6), "latent_dim": 32, "filters": (32, 64, 128, 256, 512, 512), "kernels": (3,) * 6}
# The evaluations worth a worker process (>= 20 s inline), in the order the suite reaches them.  A test asks for its case
# with oracle_pool.fetch(function name, **kwargs): a case listed here is computed ahead, any other one inline.
HEAVY = [
    ("f32_case", dict(arch_kw=FULL, B=256, seed=2, data_seed=5, sigma_bias=0.0, train_decoder=True, f32_floor=True)),
    ("f32_case", dict(arch_kw=FULL, B=256, seed=2, data_seed=5, sigma_bias=0.3, train_decoder=True, f32_floor=True)),
    ("f32_case", dict(arch_kw=FULL, B=256, seed=3, data_seed=6, sigma_bias=0.3, train_decoder=False, f32_floor=True)),
    ("f32_case", dict(arch_kw=DEEP, B=64, seed=21, data_seed=None, sigma_bias=0.3, train_decoder=True, f32_floor=True)),
    ("bf16_case", dict(arch_kw=FULL, B=256, seed=5, data_seed=9, train_decoder=True)),
    ("bf16_case", dict(arch_kw=FULL, B=64, seed=3, data_seed=6, train_decoder=True)),
    ("bf16_case", dict(arch_kw=DEEP, B=64, seed=23, data_seed=None, train_decoder=True)),
    ("bf16_case", dict(arch_kw=dict(FULL, input_shape=(59, 59, 10)), B=64, seed=97, data_seed=None, train_decoder=True)),
]
