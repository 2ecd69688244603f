"""BASELINE configs[4]: deblend() over cutouts of a field_img_2.npy-style scene, 8192 stamps per network call.

The reference cuts stamps at detected positions out of a (1, 259, 259, 6) float64 field and calls the network ONCE on all of
them (deblend/field_deblender.py:265-274 -> deblend_cutout/deblender.py:18).  Here a 259 x 259 x 6 field (the file given
with --field, e.g. the reference's field_img_2.npy, or a synthetic scene of Gaussian blobs with the same shape and
value range) is tiled into a larger scene, N windows are cut at random integer positions on the GPU
(dv_scene_extract) and go through deblend() in chunks, as DeblendField does - float64 cutouts on the host, the float32
cast inside the library.  With several ranks each takes a contiguous range of the windows (deblend_sharded, no
collective on the data path).

    python tools/field_cutouts.py [--n 1000000] [--chunk 8192] [--dtype 0|1] [--field path.npy] [--tiles 8]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_field(size=259, nb=6, seed=0):
    """A field_img_2.npy-style scene: ~60 elliptical Gaussian blobs with the DC2 band ratios plus per-band noise."""
    from debvader_amd.data import _NOISE, _SED

    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    img = np.zeros((size, size))
    for _ in range(60):
        cx, cy = rng.uniform(0, size, 2)
        s, q, th = rng.uniform(1.5, 4.0), rng.uniform(0.5, 1.0), rng.uniform(0, np.pi)
        peak = np.exp(rng.uniform(np.log(0.5), np.log(15.0)))
        dx, dy = xx - cx, yy - cy
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        img += peak * np.exp(-0.5 * (u * u / (s * s) + v * v / (s * s * q * q)))
    return img[..., None] * _SED[:nb] + rng.normal(size=(size, size, nb)) * _NOISE[:nb]


def run(ctx=None, n_cutouts=1_000_000, chunk=8192, dtype=0, field=None, tiles=8, rank=0, world=1, seed=0):
    from debvader_amd import engine as E
    from debvader_amd.parallel import shard_range

    ctx = ctx or E.default_context()
    if field is None:
        field = synthetic_field()
    field = np.asarray(field, np.float64).reshape(field.shape[-3:])
    scene = np.ascontiguousarray(np.tile(field, (tiles, tiles, 1)))
    F, cs = scene.shape[0], 59
    starts = np.random.default_rng(seed).integers(0, F - cs + 1, size=(n_cutouts, 2)).astype(np.int32)
    lo, hi = shard_range(n_cutouts, rank, world)
    eng = E.Engine(E.make_config(max_batch=chunk, dtype=dtype), ctx)
    eng.init(seed=0)
    out = {"loc": np.empty((chunk, cs, cs, 6), np.float32), "scale": np.empty((chunk, cs, cs, 6), np.float32)}
    # warm-up: kernel attributes, pinned staging, page faults
    cut = ctx.scene_extract(scene, starts[lo:lo + min(chunk, hi - lo)], cs)
    eng.infer(cut, seed=1, want=("loc", "scale"))
    t_extract = t_net = 0.0
    checksum = 0.0
    t0 = time.perf_counter()
    for b in range(lo, hi, chunk):
        e = min(hi, b + chunk)
        ta = time.perf_counter()
        cut = ctx.scene_extract(scene, starts[b:e], cs)                    # float64, as extract_cutouts returns
        tb = time.perf_counter()
        o = {k: v[:e - b] for k, v in out.items()}
        eng.infer(cut, seed=2 + b, want=("loc", "scale"), out=o)           # deblend(): mean and stddev of every stamp
        tc = time.perf_counter()
        t_extract += tb - ta
        t_net += tc - tb
        checksum += float(o["loc"][::97, 29, 29, 2].sum())
    total = time.perf_counter() - t0
    # the same forward with the stamps resident in HBM (no host copies): one chunk uploaded once, evaluated repeatedly
    nres = min(chunk, hi - lo)
    x32 = cut[:nres].astype(np.float32)
    eng.upload(1, x32, x32)
    reps = 5
    eng.eval_step(1, first=0, B=nres, seed=3)
    ctx.sync()
    t1 = time.perf_counter()
    for r in range(reps):
        eng.eval_step(1, first=0, B=nres, seed=4 + r)
    ctx.sync()
    t_res = (time.perf_counter() - t1) / reps
    eng.close()
    n = hi - lo
    return {
        "workload": f"BASELINE configs[4] per GPU: deblend() over {n} cutouts (59x59x6) of a {F}x{F}x6 scene tiled from a "
                    f"259x259x6 field, {chunk} per network call, {'bf16' if dtype else 'fp32'} engine",
        "value": n / total, "unit": "stamps/s", "dtype": "bf16" if dtype else "f32", "n_cutouts": n, "chunk": chunk,
        "includes": "cutout gather on the GPU with float64 D2H, float64 -> float32 staging, H2D, forward, D2H of mean and stddev",
        "deblend_only_stamps_per_s": n / t_net, "extract_only_stamps_per_s": n / t_extract,
        "resident_forward_stamps_per_s": nres / t_res,
        "resident_forward_note": "same forward + head on a chunk already in HBM (dv_eval_step), no host copies",
        "checksum": checksum,
    }


def main():
    import argparse
    import json

    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--field", default=None)
    ap.add_argument("--tiles", type=int, default=8)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from debvader_amd import engine as E

    ctx = E.Context(int(os.environ.get("LOCAL_RANK", "0")), 0, 1, None)      # no collective: every rank is on its own
    field = np.load(a.field) if a.field else None
    res = run(ctx, a.n, a.chunk, a.dtype, field, a.tiles, rank, world)
    res["rank"], res["world"] = rank, world
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
