"""Layer-by-layer comparison of the bf16 engine's forward with the bf16-storage oracle (development aid)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae_oracle as vo
from oracle import vae_oracle_bf16 as vb
from debvader_amd import engine as E
from tools.bf16_probe import case, relmax

arch = vo.Arch(input_shape=(13, 13, 4), latent_dim=8, filters=(16, 32), kernels=(3, 3))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p, x, y, eps = case(arch, B, 0)
cfg = E.make_config(arch.input_shape, arch.latent_dim, tuple(arch.filters), tuple(arch.kernels), max_batch=B, dtype=1)
eng = E.Engine(cfg)
eng.set_params(p)
eng.optimizer_reset(1e-4)
eng.upload(0, x, y)
eng.keep_outputs(True)
x64, y64, e64 = x.astype(np.float64), y.astype(np.float64), eps.astype(np.float64)
cb = vb.forward(arch, p, x64, e64, training=True)
out = eng.grad_step(0, first=0, B=B, eps=eps)
H = arch.input_shape[0]
xn = eng.activation("xn", (B, H, H, 16))
ref = cb["enc_in0"]
print("xn", relmax(xn[..., :ref.shape[-1]], ref), "pad max", np.abs(xn[..., ref.shape[-1]:]).max())
sizes = arch.enc_sizes
for j in range(2 * len(arch.filters)):
    hout = sizes[j // 2 + 1] if j % 2 else sizes[j // 2]
    cout = arch.filters[j // 2]
    u = eng.activation(f"enc_u{j}", (B, hout, hout, cout))
    print(f"enc_u{j}", relmax(u, cb[f"enc_u{j}"]), u.shape)
    if j == 0:
        d = np.abs(u - cb["enc_u0"])
        print("  per-channel max err", d.max(axis=(0, 1, 2)).round(3))
        print("  per-row max err", d.max(axis=(0, 2, 3)).round(3))
        print("  per-stamp max err", d.max(axis=(1, 2, 3)).round(3))
print("t", relmax(eng.activation("t", (B, arch.params_size)), cb["t"]))
size = arch.w0
for j in range(2 * len(arch.filters)):
    if j % 2 == 0:
        size *= 2
    cout = arch.filters[len(arch.filters) - 1 - j // 2]
    u = eng.activation(f"dec_u{j}", (B, size, size, cout))
    print(f"dec_u{j}", relmax(u, cb[f"dec_u{j}"]), u.shape)
eng.close()
