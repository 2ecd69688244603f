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


def run(ctx=None, n_cutouts=1_000_000, chunk=8192, dtype=0, field=None, tiles=8, rank=0, world=1, seed=0, fused=False,
        calls=None, stream=False, on_device=False):
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
    if on_device:
        return _run_on_device(ctx, eng, scene, starts, lo, hi, chunk, dtype, F, cs)
    if stream:
        return _run_stream(ctx, eng, scene, starts, lo, hi, chunk, dtype, F, cs)
    per_call = chunk * (calls or (4 if fused else 1))       # stamps per engine call (fused: the field is uploaded per call)
    out = {"loc": np.empty((per_call, cs, cs, 6), np.float32), "scale": np.empty((per_call, cs, cs, 6), np.float32)}
    # warm-up: kernel attributes, pinned staging, page faults
    cut = ctx.scene_extract(scene, starts[lo:lo + min(chunk, hi - lo)], cs)
    eng.infer(cut, seed=1, want=("loc", "scale"))
    if fused:                                     # the pinned transfer ring of the timed calls' chunk size
        eng.infer_cutouts(scene, starts[lo:lo + min(2 * chunk, hi - lo)], seed=1, want=("loc", "scale"),
                          out={k: v[:min(2 * chunk, hi - lo)] for k, v in out.items()})
    t_extract = t_net = 0.0
    checksum = 0.0
    t0 = time.perf_counter()
    for b in range(lo, hi, per_call):
        e = min(hi, b + per_call)
        o = {k: v[:e - b] for k, v in out.items()}
        ta = time.perf_counter()
        if fused:
            # deblend_field_cutouts: gather + float32 cast on the GPU, stamps never visit the host (dv_infer_cutouts)
            tb = ta
            eng.infer_cutouts(scene, starts[b:e], seed=2 + b, want=("loc", "scale"), out=o)
        else:
            cut = ctx.scene_extract(scene, starts[b:e], cs)                # float64, as extract_cutouts returns
            tb = time.perf_counter()
            eng.infer(cut, seed=2 + b, want=("loc", "scale"), out=o)       # deblend(): mean and stddev of every stamp
        tc = time.perf_counter()
        t_extract += tb - ta
        t_net += tc - tb
        checksum += float(o["loc"][::97, 29, 29, 2].sum())
    total = time.perf_counter() - t0
    if fused:
        cut = ctx.scene_extract(scene, starts[lo:lo + min(chunk, hi - lo)], cs)
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
        "includes": ("field H2D once per call, cutout gather + float32 cast on the GPU, forward, D2H of mean and stddev "
                     f"(deblend_field_cutouts, {per_call} cutouts per call)") if fused else
                    "cutout gather on the GPU with float64 D2H, float64 -> float32 staging, H2D, forward, D2H of mean and stddev",
        "deblend_only_stamps_per_s": n / t_net, "extract_only_stamps_per_s": (n / t_extract) if t_extract > 0 else None,
        "resident_forward_stamps_per_s": nres / t_res,
        "resident_forward_note": "same forward + head on a chunk already in HBM (dv_eval_step), no host copies",
        "checksum": checksum,
    }


def run_drop_in(ctx=None, n_cutouts=131072, chunk=8192, dtype=0, field=None, tiles=8, per_call=32768, seed=0):
    """The reference's OWN call sequence, nothing engine-specific in it: `DeblendField(net, field).deblend_field(distances)`
    (deblend/field_deblender.py:219-383 with its defaults) on a net from create_model_vae, `per_call` galaxies per call,
    the recarray with float64 cutout_images and float32 mean / stddev stamps per galaxy coming back every time.  Since round 5
    that is one engine call per deblend_field (dv_infer_cutouts_keep); before, extract_cutouts -> deblend with a float64 D2H,
    a host cast and an H2D per chunk."""
    from debvader_amd import engine as E
    from debvader_amd.deblend.field_deblender import DeblendField
    from debvader_amd.model import model

    ctx = ctx or E.default_context()
    if field is None:
        field = synthetic_field()
    field = np.asarray(field, np.float64).reshape(field.shape[-3:])
    scene = np.ascontiguousarray(np.tile(field, (tiles, tiles, 1)))
    F, cs = scene.shape[0], 59
    starts = np.random.default_rng(seed).integers(0, F - cs + 1, size=(n_cutouts, 2))
    dist = (starts + cs // 2 - F // 2).astype(np.float64)        # extraction.py:26-30 read backwards: start -> distance to centre
    net, _, _, _ = model.create_model_vae((cs, cs, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=chunk, ctx=ctx, seed=0,
                                          dtype="bf16" if dtype else "float32")
    db = DeblendField(net, scene[None])
    eng = net._core.engine
    t_eng = [0.0]
    real = eng.infer_cutouts_keep

    def timed(*a, **k):
        t = time.perf_counter()
        r = real(*a, **k)
        t_eng[0] += time.perf_counter() - t
        return r

    eng.infer_cutouts_keep = timed
    # warm-up with a call of the size that is timed: kernel attributes, the pinned transfer ring, and the result arrays'
    # blocks (engine._HostPool) - a smaller warm-up leaves smaller blocks behind whose ageing out of the pool (an munmap of
    # 2.7 GB, ~0.1 s) then falls into the timed calls (tools/probes/drop_in_calls.py: 258, 258, 258, 383, 258, 258 ms per call)
    nw = min(per_call, n_cutouts)
    res = db.deblend_field(dist[:nw])
    assert len(res) == nw
    del res
    t_eng[0] = 0.0
    checksum, n_pass = 0.0, 0
    t0 = time.perf_counter()
    for b in range(0, n_cutouts, per_call):
        res = db.deblend_field(dist[b:b + per_call])
        checksum += float(res["output_images_mean"][0][29, 29, 2]) + float(res["cutout_images"][-1][29, 29, 2])
        n_pass += int(np.sum(res["passed_cuts"]))
        del res
    total = time.perf_counter() - t0
    # ... and ONE call of the same size with the result-array pool switched off (what DV_HOST_POOL_GB=0 gives: every
    # result array fresh from np.empty, page-faulted while the copy threads fill it, unmapped when it is dropped)
    pool_cap, E._host_pool.cap = E._host_pool.cap, 0
    E.host_pool_clear()
    try:
        res = db.deblend_field(dist[:per_call])
        del res
        db.res_deblend = None
        t2 = time.perf_counter()
        res = db.deblend_field(dist[:per_call])
        checksum += float(res["output_images_mean"][0][29, 29, 2])
        del res
        db.res_deblend = None
        t_nopool = time.perf_counter() - t2
    finally:
        E._host_pool.cap = pool_cap
    # the same forward with the stamps resident in HBM (no host copies)
    nres = min(chunk, n_cutouts)
    x32 = ctx.scene_extract(scene, starts[:nres], cs).astype(np.float32)
    eng.upload(1, x32, x32)
    eng.eval_step(1, first=0, B=nres, seed=3)
    ctx.sync()
    reps = 5
    t1 = time.perf_counter()
    for r in range(reps):
        eng.eval_step(1, first=0, B=nres, seed=4 + r)
    ctx.sync()
    t_res = (time.perf_counter() - t1) / reps
    eng.close()
    n = n_cutouts
    return {
        "workload": f"BASELINE configs[4] per GPU, the reference's own call sequence: DeblendField(net, field).deblend_field("
                    f"distances) over {n} galaxies of a {F}x{F}x6 scene tiled from a 259x259x6 field, {per_call} per call, "
                    f"{chunk} per network pass, {'bf16' if dtype else 'fp32'} engine",
        "value": n / total, "unit": "stamps/s", "dtype": "bf16" if dtype else "f32", "n_cutouts": n, "chunk": chunk,
        "per_call": per_call,
        "includes": "per call: field H2D, cutout gather + float32 cast on the GPU, forward, D2H of mean and stddev through the "
                    "pinned ring into result arrays RECYCLED from the previous call (engine._HostPool: the loop drops each "
                    "recarray before the next call, so its blocks are handed out again), the float64 cutout_images assembled "
                    "on the host from the field beside the forward passes (dv_infer_cutouts_keep), the centre-MSE quality cut "
                    "and the pandas recarray of the reference",
        "definition_changed_in": "r05: until r04 this entry timed extract_cutouts -> deblend (tools/field_cutouts.py::cutouts_run, "
                                 "43.7 k stamps/s there); since r05 it times DeblendField.deblend_field per 32768 galaxies - "
                                 "not like for like",
        "steady_state": "one untimed call of the same size first: every timed call reuses the previous call's result blocks",
        "value_without_result_pool": min(per_call, n) / t_nopool,
        "value_without_result_pool_note": "one call of the same size with DV_HOST_POOL_GB=0 semantics: fresh np.empty result arrays",
        "engine_call_stamps_per_s": n / t_eng[0] if t_eng[0] > 0 else None,
        "python_side_s": total - t_eng[0],
        "host_bytes_per_stamp": cs * cs * 6 * (8 + 4 + 4), "link_bytes_per_stamp": cs * cs * 6 * 8,
        "resident_forward_stamps_per_s": nres / t_res,
        "resident_forward_note": "same forward + head on a chunk already in HBM (dv_eval_step), no host copies",
        "passed_cuts": n_pass, "checksum": checksum,
    }


def _run_on_device(ctx, eng, scene, starts, lo, hi, chunk, dtype, F, cs):
    """DeblendField.deblend_field(on_device=True) + get_predicted_field / get_residual_field: cutout gather, network and the
    compositing of every stamp's mean and stddev into the field-sized results in ONE engine call (dv_infer_cutouts_composite);
    only three F x F x 6 float64 fields and one float64 per stamp come back."""
    places = starts.astype(np.int64)               # a stamp is put back where it was cut (integer positions)
    eng.infer_cutouts_composite(scene, starts[lo:lo + min(2 * chunk, hi - lo)], places[lo:lo + min(2 * chunk, hi - lo)], seed=1)
    t0 = time.perf_counter()
    out = eng.infer_cutouts_composite(scene, starts[lo:hi], places[lo:hi], seed=2)
    total = time.perf_counter() - t0
    n = hi - lo
    # the same forward with the stamps resident in HBM (no gather, no compositing, no copies)
    nres = min(chunk, n)
    x32 = ctx.scene_extract(scene, starts[lo:lo + nres], cs).astype(np.float32)
    eng.upload(1, x32, x32)
    eng.eval_step(1, first=0, B=nres, seed=3)
    ctx.sync()
    reps = 5
    t1 = time.perf_counter()
    for r in range(reps):
        eng.eval_step(1, first=0, B=nres, seed=4 + r)
    ctx.sync()
    t_res = (time.perf_counter() - t1) / reps
    eng.close()
    return {
        "workload": f"BASELINE configs[4] per GPU: DeblendField.deblend_field(on_device=True) + get_predicted_field + "
                    f"get_residual_field over {n} cutouts (59x59x6) of a {F}x{F}x6 scene, {chunk} per network call, "
                    f"{'bf16' if dtype else 'fp32'} engine",
        "value": n / total, "unit": "stamps/s", "dtype": "bf16" if dtype else "f32", "n_cutouts": n, "chunk": chunk,
        "includes": "field H2D once, cutout gather + float32 cast on the GPU, forward, mean / stddev / residual fields composited "
                    "on the GPU in object order (float64), centre MSE per stamp; D2H of three F x F x 6 fields and N doubles",
        "host_bytes_out": 3 * F * F * 6 * 8 + 8 * n,
        "resident_forward_stamps_per_s": nres / t_res,
        "fraction_of_resident_forward": (n / total) / (nres / t_res),
        "checksum": float(out["mean_field"][::37, ::41, 2].sum()),
        "mean_of_mse_center": float(out["mse_center"].mean()),
    }


def _run_stream(ctx, eng, scene, starts, lo, hi, chunk, dtype, F, cs):
    """deblend_field_cutouts(on_chunk=...): one engine call for the whole range, results consumed chunk by chunk from the
    pinned transfer ring (here: a checksum over every chunk and running sums of mean and stddev of the centre pixel)."""
    state = {"checksum": 0.0, "n": 0, "sum_std": 0.0}

    def consume(first, mean, std):
        state["checksum"] += float(mean[::97, 29, 29, 2].sum())
        state["sum_std"] += float(std[:, 29, 29, 2].sum())
        state["n"] += mean.shape[0]

    # warm-up with the pipeline geometry of the timed call (two full chunks: the pinned transfer ring, ~8 GB at 8192
    # stamps per chunk, is allocated once per model and chunk size)
    eng.infer_cutouts_stream(scene, starts[lo:lo + min(2 * chunk, hi - lo)], lambda *a: None, seed=1)
    t0 = time.perf_counter()
    eng.infer_cutouts_stream(scene, starts[lo:hi], consume, seed=2)
    total = time.perf_counter() - t0
    eng.close()
    n = hi - lo
    assert state["n"] == n
    return {
        "workload": f"BASELINE configs[4] per GPU: deblend() over {n} cutouts (59x59x6) of a {F}x{F}x6 scene tiled from a "
                    f"259x259x6 field, {chunk} per network call, {'bf16' if dtype else 'fp32'} engine, streaming consumer",
        "value": n / total, "unit": "stamps/s", "dtype": "bf16" if dtype else "f32", "n_cutouts": n, "chunk": chunk,
        "includes": "field H2D once, cutout gather + float32 cast on the GPU, forward, D2H of mean and stddev of every stamp "
                    "into the pinned ring, consumed in place (deblend_field_cutouts(on_chunk=...), dv_infer_cutouts_stream)",
        # what bounds this entry: every stamp's mean and stddev cross PCIe (2 x 59*59*6 float32 = 167 088 bytes per stamp)
        "d2h_bytes_per_stamp": 2 * cs * cs * 6 * 4, "d2h_gbs": n * 2 * cs * cs * 6 * 4 / total / 1e9,
        "bound": "host link: d2h_gbs is this box's device-to-host rate (pinned ring, two copy streams); the GPU side alone "
                 "runs at secondary.deblend_cutouts' resident_forward_stamps_per_s",
        "checksum": state["checksum"], "mean_stddev_centre_pixel": state["sum_std"] / n,
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
    ap.add_argument("--fused", action="store_true", help="deblend_field_cutouts: gather on the GPU, no host round trip")
    ap.add_argument("--stream", action="store_true", help="deblend_field_cutouts(on_chunk=...): results consumed per chunk")
    ap.add_argument("--on-device", action="store_true", help="compositing on the GPU behind the forward passes: only fields come back")
    ap.add_argument("--drop-in", action="store_true", help="DeblendField.deblend_field(distances), the reference's call sequence")
    ap.add_argument("--per-call", type=int, default=32768)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from debvader_amd import engine as E

    ctx = E.Context(int(os.environ.get("LOCAL_RANK", "0")), 0, 1, None)      # no collective: every rank is on its own
    field = np.load(a.field) if a.field else None
    if a.drop_in:
        print(json.dumps(run_drop_in(ctx, a.n, a.chunk, a.dtype, field, a.tiles, a.per_call)), flush=True)
        return
    res = run(ctx, a.n, a.chunk, a.dtype, field, a.tiles, rank, world, fused=a.fused, stream=a.stream, on_device=a.on_device)
    res["rank"], res["world"] = rank, world
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
