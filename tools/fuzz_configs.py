"""Corner-case sweep of both engines: architectures / batch sizes the tests do not pin, one train step + one inference
each; prints what is refused (with the library's message) and fails on anything that is accepted but not finite."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd import engine as E
from debvader_amd._lib import DvError

rng = np.random.default_rng(0)
cases = []
for size, bands, filters, latent in [(32, 2, (16, 32), 8), (45, 4, (32, 64, 128), 16), (59, 6, (32, 32, 32, 32), 32),
                                     (64, 6, (16, 32, 64), 8), (20, 4, (32, 64), 8), (59, 6, (32, 64, 128, 256), 16),
                                     (16, 2, (32,), 8), (59, 6, (64, 64, 64, 64), 32), (40, 4, (32, 96), 16),
                                     (33, 6, (32, 64, 64), 32), (24, 2, (8, 16), 8), (59, 6, (16, 16, 16, 16), 8),
                                     (27, 4, (24, 48), 8), (30, 2, (12, 20), 8), (36, 6, (16, 96), 16),
                                     (48, 6, (32, 160, 224), 32), (100, 2, (32, 64, 96, 128), 64), (21, 6, (32, 64), 32),
                                     (10, 6, (16, 32), 8), (8, 2, (32, 32, 32), 8), (59, 6, (32, 64, 128, 256, 256), 32),
                                     (9, 4, (32,), 8)]:
    for B in (1, 7, 64, 100, 256, 300):
        cases.append((size, bands, filters, latent, B, (3,) * len(filters)))
# round 4: any band count, kernel sizes 1 .. 5 per level (general gather-GEMM / tiled weight gradient); round 5: the bf16
# engine takes them too, and both take latent sizes that are not multiples of 4; the fp32 engine 8 .. 15 bands
for size, bands, filters, latent, kernels in [(20, 5, (32, 64), 8, (5, 5)), (32, 1, (16, 32), 8, (3, 5)),
                                              (27, 3, (24, 48), 8, (1, 3)), (45, 7, (32, 64, 128), 16, (5, 3, 1)),
                                              (59, 6, (32, 64, 128, 256), 32, (5, 3, 5, 3)), (16, 2, (32,), 8, (4,)),
                                              (30, 6, (16, 32), 8, (2, 4)), (27, 10, (16, 32), 10, (3, 3)),
                                              (20, 4, (32, 64), 5, (5, 3)), (59, 15, (32, 64), 30, (3, 3))]:
    for B in (1, 7, 64, 100):
        cases.append((size, bands, filters, latent, B, kernels))
bad = 0
for dtype in (0, 1):
    for size, bands, filters, latent, B, kernels in cases:
        tag = f"dtype {dtype} size {size} bands {bands} filters {filters} kernels {kernels} latent {latent} B {B}"
        try:
            cfg = E.make_config((size, size, bands), latent, filters, kernels, max_batch=B, dtype=dtype)
            eng = E.Engine(cfg)
        except (DvError, ValueError) as e:
            if B == 1:
                print("refused:", tag, "|", str(e)[:110])
            continue
        try:
            eng.init(seed=1)
            eng.optimizer_reset(1e-4)
            x = rng.normal(0, 0.4, size=(B, size, size, bands)).astype(np.float32)
            y = np.abs(x) * 0.5
            eng.upload(0, x, y)
            o1 = eng.train_step(0, first=0, B=B, seed=1)
            o2 = eng.train_step(0, first=0, B=B, seed=2)
            r = eng.infer(x[: min(B, 9)], seed=3)
            ok = np.isfinite([o1["loss"], o2["loss"]]).all() and np.isfinite(r["loc"]).all() and (r["scale"] > 0).all()
            # stage 2 (decoder frozen), evaluation, chunked inference beyond max_batch, Monte-Carlo statistics
            eng.set_trainable(True, False)
            eng.optimizer_reset(1e-4)
            o3 = eng.train_step(0, first=0, B=B, seed=4)
            eng.upload(1, x[: max(1, B // 2)], y[: max(1, B // 2)])
            o4 = eng.eval_step(1, first=0, B=max(1, B // 2), seed=5)
            big = np.concatenate([x, x, x[:3]]) if B <= 64 else x
            r2 = eng.infer(big, seed=6, want=("loc", "mu", "zstd"))
            mean, std = eng.infer_mc(x[: min(B, 5)], nsamples=3, seed=7)
            ok = ok and np.isfinite([o3["loss"], o4["loss"]]).all() and np.isfinite(r2["loc"]).all() and \
                np.isfinite(r2["zstd"]).all() and np.isfinite(mean).all() and np.isfinite(std).all()
            if not ok:
                bad += 1
                print("NOT FINITE:", tag, o1, o2)
        except DvError as e:
            bad += 1
            print("FAILED:", tag, "|", str(e)[:160])
        finally:
            eng.close()
print("done,", bad, "problem(s)")
sys.exit(1 if bad else 0)
