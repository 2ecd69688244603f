"""Manual helper (not collected by pytest): prints per-tensor GPU-vs-oracle errors without stopping."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import vae_oracle as vo
from tests.test_gpu_parity import small_arch, _case, _engine, _relmax


def report(arch, B, seed, data=None):
    p, x, y, eps = _case(arch, B, seed, data)
    eng = _engine(arch, max_batch=B)
    eng.set_params(p)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    c = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=True)
    ref = vo.losses(arch, c, y.astype(np.float64))
    g = vo.backward(arch, p, c, y.astype(np.float64))
    out = eng.grad_step(0, first=0, B=B, eps=eps)
    H, W, C = arch.input_shape
    L = len(arch.filters)
    names = [("xn", None)]
    def act(name, ref_arr):
        v = eng.activation(name, ref_arr.shape)
        print(f"  act {name:10s} relmax {_relmax(v, ref_arr):.3e}")
    xn = np.zeros((B, H, W, 8)); xn[..., :C] = c["enc_in0"]
    act("xn", xn)
    for j in range(2 * L):
        act(f"enc_u{j}", c[f"enc_u{j}"])
    act("t", c["t"]); act("z", c["z"]); act("kl", c["kl"])
    for j in range(2 * L):
        act(f"dec_u{j}", c[f"dec_u{j}"])
    act("head_pre", c["head_pre"]); act("loc", c["loc"]); act("scale", c["scale"])
    for k in ("loss", "nll_mean", "kl_reg", "mse"):
        print(f"  scalar {k:9s} gpu {out[k]:.8e} ref {ref[k]:.8e} rel {abs(out[k]-ref[k])/abs(ref[k]):.2e}")
    for name, _, tr in arch.param_specs():
        if name in g:
            print(f"  grad {name:24s} relmax {_relmax(eng.get_grad(name), g[name]):.3e}  max|g| {np.abs(g[name]).max():.3e}")
    eng.close()


if __name__ == "__main__":
    print("small arch B=5"); report(small_arch(), 5, 0)
    if len(sys.argv) > 1:
        from debvader_amd.data import synthetic_stamps
        x, y = synthetic_stamps(4, seed=5)
        print("full arch B=4"); report(vo.Arch(), 4, 2, (x, y))
