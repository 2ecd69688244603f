"""Repeats the same B=256 gradient step (or, with "train", six queued train steps) on fresh engines and reports any
gradient (parameter) that is not bit-identical to the first run: a race detector for the multi-stream step.
GPU only.  usage: determinism_probe.py [runs] [train] [dtype 0|1] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
B = 256
rng = np.random.default_rng(7)
NROWS = 2 * B if len(sys.argv) > 2 else B
x = rng.normal(size=(NROWS, 59, 59, 6)).astype(np.float32)
y = rng.normal(size=(NROWS, 59, 59, 6)).astype(np.float32)
eps = np.random.default_rng(1).normal(size=(B, 32)).astype(np.float32)
ref = None
bad = 0
for r in range(runs):
    eng = E.Engine(E.make_config(max_batch=B, dtype=dtype))
    eng.init(seed=5)
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    names = [n for n, _, trainable in eng.specs if trainable]
    if len(sys.argv) > 2 and sys.argv[2] == "train":
        o = eng.train_steps(0, 0, B, nsteps, seed=10)
        out = {"loss": o["loss"]}
        g = {n: eng.get_param(n) for n in names}
    else:
        out = eng.grad_step(0, first=0, B=B, eps=eps)
        g = {n: eng.get_grad(n) for n in names}
    eng.close()
    if ref is None:
        ref = (out, g)
        continue
    diffs = [(n, float(np.abs(g[n] - ref[1][n]).max()), int((g[n] != ref[1][n]).sum())) for n in names
             if not np.array_equal(g[n], ref[1][n])]
    if diffs or out != ref[0]:
        bad += 1
        print(f"run {r}: loss {out['loss']!r} vs {ref[0]['loss']!r}; differing gradients:")
        for n, d, c in diffs:
            print(f"   {n}: max |diff| {d:.3e} in {c} elements of {g[n].size}")
print(f"dtype {dtype}, {'train x' + str(nsteps) if len(sys.argv) > 2 and sys.argv[2] == 'train' else 'gradient step'}: {bad} of {runs - 1} repeat runs differ")
