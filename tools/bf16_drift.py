"""Where do the bf16 and the fp32 engine's training curves separate, and why?  (VERDICT r2 "Next round" 1.)

Three runs from the same initialisation on the same batches, each in its own process:
  f32     the fp32 engine as shipped
  f32alt  the fp32 engine with other kernels for the same layers (dv_debug_general_kernels(1): the general gather-GEMM
          and direct weight-gradient kernels instead of the Winograd / strip / fused stride-2 forms; one forward lane):
          the SAME arithmetic in another summation order
  bf16    the bf16-storage engine
Per step: loss / NLL / KL, the smallest sigma of the batch, how many pixels sit on the 1e-4 floor (model.py:154-159),
the largest |y - mu| / sigma, and every few steps the norm of the parameter update per group; validation loss on held-out
stamps at a few checkpoints.  Output: one JSON per run + a summary table (gpurun_out/ by default).

    python tools/bf16_drift.py [--steps 300] [--batch 64] [--shift 0.3] [--out gpurun_out/drift]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "f32": (0, {}),
    "f32alt": (0, {"DV_NO_FWD_SPLIT": "1"}),
    "bf16": (1, {}),
}


def run_variant(name, a):
    import numpy as np
    from debvader_amd import engine as E
    from debvader_amd.data import synthetic_stamps

    dtype = VARIANTS[name][0]
    if name == "f32alt":
        from debvader_amd._lib import check
        from tests import debug_lib
        lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
        check(lib.dv_debug_general_kernels(1))
    B, steps = a.batch, a.steps
    ntrain, nval = a.ntrain, a.nval
    x, y = synthetic_stamps(ntrain + nval, seed=21)
    if a.label_noise:
        # noisy labels: the optimal sigma is the noise level, far above the 1e-4 floor (with noise-free labels the
        # optimum is sigma -> floor wherever the label is exactly zero)
        from debvader_amd.data import _NOISE
        y = y + (np.random.default_rng(77).normal(size=y.shape) * _NOISE[:6] * a.label_noise).astype(np.float32)
    xv, yv = x[ntrain:], y[ntrain:]
    x, y = x[:ntrain], y[:ntrain]
    eng = E.Engine(E.make_config(max_batch=B, dtype=dtype, sigma_floor=a.sigma_floor))
    eng.init(seed=5)
    if a.shift:
        hb = eng.get_param("dec/head/bias")
        hb[6:] += a.shift
        eng.set_param("dec/head/bias", hb)
    eng.optimizer_reset(a.lr)
    eng.upload(0, x, y)
    eng.upload(1, xv, yv)
    eng.keep_outputs(True)
    groups = {"enc_conv": [n for n, _, t in eng.specs if t and n.startswith("enc/conv")],
              "enc_dense": ["enc/dense/kernel", "enc/dense/bias"],
              "dec_dense": [n for n, _, t in eng.specs if t and n.startswith("dec/dense")],
              "dec_convt": [n for n, _, t in eng.specs if t and n.startswith("dec/convt")],
              "alpha": [n for n, _, t in eng.specs if t and n.endswith("/alpha")],
              "head": ["dec/head/kernel", "dec/head/bias"]}
    rec = dict(variant=name, dtype=dtype, batch=B, steps=steps, lr=a.lr, shift=a.shift, step=[], val=[], upd=[])
    floor = a.sigma_floor * (1 + 1e-5)
    prev = None
    nb = ntrain // B

    def val_loss():
        tot = {"loss": 0.0, "nll_mean": 0.0, "kl_reg": 0.0}
        n = 0
        for k in range(nval // B):
            o = eng.eval_step(1, first=k * B, B=B, seed=7000 + k)
            for key in tot:
                tot[key] += o[key]
            n += 1
        return {k: v / n for k, v in tot.items()}

    for s in range(steps):
        first = (s % nb) * B
        o = eng.train_step(0, first=first, B=B, seed=100 + s)
        sc = eng.activation("scale", (B, 59, 59, 6))
        lo = eng.activation("loc", (B, 59, 59, 6))
        r = np.abs(y[first:first + B] - lo) / sc
        rec["step"].append(dict(s=s, loss=o["loss"], nll=o["nll_mean"], kl=o["kl_reg"], mse=o["mse"],
                                sig_min=float(sc.min()), n_floor=int((sc <= floor).sum()),
                                n_sig_lt_1e3=int((sc < 1e-3).sum()), r_max=float(r.max()), n_r_gt_30=int((r > 30).sum())))
        if s % a.upd_every == 0 or s == steps - 1:
            cur = {g: np.concatenate([eng.get_param(n).ravel() for n in names]) for g, names in groups.items()}
            if prev is not None:
                rec["upd"].append(dict(s=s, **{g: float(np.linalg.norm(cur[g] - prev[g])) for g in groups}))
            prev = cur
        if (s + 1) in a.val_at or s == steps - 1:
            v = val_loss()
            rec["val"].append(dict(s=s + 1, **v))
            print(f"[{name}] step {s + 1}: train loss {o['loss']:.5f}  val loss {v['loss']:.5f}  sig_min {sc.min():.3e} "
                  f"floor px {int((sc <= floor).sum())}", flush=True)
    eng.close()
    with open(a.out + f"_{name}.json", "w") as f:
        json.dump(rec, f)


def summary(a):
    import numpy as np

    recs = {}
    for name in VARIANTS:
        p = a.out + f"_{name}.json"
        if os.path.exists(p):
            recs[name] = json.load(open(p))
    lines = []
    names = list(recs)
    L = {n: np.array([s["loss"] for s in recs[n]["step"]]) for n in names}
    F = {n: np.array([s["n_floor"] for s in recs[n]["step"]]) for n in names}
    R = {n: np.array([s["r_max"] for s in recs[n]["step"]]) for n in names}
    lines.append("step " + " ".join(f"{n:>11s} floor   rmax" for n in names))
    steps = len(next(iter(L.values())))
    for s in list(range(0, steps, max(1, steps // 40))) + [steps - 1]:
        lines.append(f"{s:4d} " + " ".join(f"{L[n][s]:11.5f} {F[n][s]:5d} {R[n][s]:6.1f}" for n in names))
    if "f32" in L:
        for n in names:
            if n == "f32":
                continue
            d = np.abs(L[n] - L["f32"])
            first = int(np.argmax(d > 0.02 * np.abs(L["f32"]).max())) if (d > 0.02 * np.abs(L["f32"]).max()).any() else -1
            lines.append(f"|{n} - f32|: max {d.max():.4f}, median {np.median(d):.5f}, first step beyond 2 % of scale: {first}")
    for n in names:
        sp = np.where(np.diff(L[n]) > 0.5)[0] + 1
        lines.append(f"{n}: loss spikes (> +0.5 in one step) at steps {sp.tolist()[:20]}; first floor pixel at step "
                     f"{int(np.argmax(F[n] > 0)) if (F[n] > 0).any() else -1}")
        lines.append(f"{n}: validation " + ", ".join(f"{v['s']}: {v['loss']:.5f}" for v in recs[n]["val"]))
        lines.append(f"{n}: mean train loss of the last 20 steps {L[n][-20:].mean():.5f}")
    txt = "\n".join(lines)
    print(txt)
    with open(a.out + "_summary.txt", "w") as f:
        f.write(txt + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ntrain", type=int, default=1024)
    ap.add_argument("--nval", type=int, default=256)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--shift", type=float, default=0.3)
    ap.add_argument("--sigma-floor", type=float, default=1e-4,
                    help="model.py:154-159 uses 1e-4; a floor of a few 1e-2 bounds the NLL's curvature (control runs)")
    ap.add_argument("--label-noise", type=float, default=0.0)
    ap.add_argument("--upd-every", type=int, default=5)
    ap.add_argument("--val-at", type=int, nargs="*", default=[25, 50, 100, 200])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "drift"))
    ap.add_argument("--variant", default=None)
    ap.add_argument("--variants", nargs="*", default=list(VARIANTS))
    a = ap.parse_args()
    if a.variant:
        run_variant(a.variant, a)
        return
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    for name in a.variants:
        env = dict(os.environ)
        env.update(VARIANTS[name][1])
        argv = [sys.executable, os.path.abspath(__file__), "--variant", name] + \
               [x for x in sys.argv[1:] if x not in ("--variant",)]
        rc = subprocess.run(argv, env=env).returncode
        if rc != 0:
            print(f"variant {name} exited with {rc}", file=sys.stderr)
            sys.exit(rc)
    summary(a)


if __name__ == "__main__":
    main()
