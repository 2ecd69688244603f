"""Whole-model check of the specialised / multi-stream paths at other batch sizes than the bench's: the same gradient
step with the default configuration and (second process, via env toggles) with the general kernels on one stream.
usage: batch_sweep.py run <B> <out.npz>  |  batch_sweep.py cmp <a.npz> <b.npz>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    from debvader_amd import engine as E
    B = int(sys.argv[2])
    from debvader_amd.data import synthetic_stamps
    # blended / isolated galaxy stamps.  BASE (env) < B: the first BASE stamps tiled - separates a batch-size
    # dependent kernel fault from an ill-conditioned stamp (at initialisation sigma sits on its 1e-4 floor wherever the
    # head's relu is at zero, and a last-bit change of the pre-activation flips that gate)
    base = int(os.environ.get("BASE", B))
    x, y = synthetic_stamps(base, seed=3)
    e0 = np.random.default_rng(3).normal(size=(base, 32)).astype(np.float32)
    reps = (B + base - 1) // base
    x, y, eps = (np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:B].copy() for a in (x, y, e0))
    eng = E.Engine(E.make_config(max_batch=B))
    eng.init(seed=5)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    out = eng.grad_step(0, first=0, B=B, eps=eps)
    g = {n.replace("/", "__"): eng.get_grad(n) for n, _, tr in eng.specs if tr}
    np.savez(sys.argv[3], loss=np.float64(out["loss"]), **g)
    print("B", B, "loss", out["loss"])
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0.0
    for k in a.files:
        if k == "loss":
            continue
        d = np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)
        worst = max(worst, d)
        if d > 2e-4:
            print("MISMATCH", k, d)
    print("loss rel diff %.2e, worst gradient rel-to-max diff %.2e" % (abs(a["loss"] - b["loss"]) / abs(b["loss"]), worst))
    sys.exit(1 if worst > 2e-4 else 0)
