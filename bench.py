#!/usr/bin/env python3
"""Headline benchmark: galaxy stamps/sec of the conv-VAE training step (fwd + ELBO + bwd + Adam),
59x59x6 stamps, batch 256 per GPU, latent 32, fp32 (BASELINE.json configs[1]); weak scaling over N GPUs.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config {1,2,3,4}]

--config selects the BASELINE configuration the line is about (default 1, the headline): 2 the same step on the bf16
engine, 3 the 128x128x6 / six-level net at 64 stamps per GPU (global 512 on 8), 4 sharded deblend() inference over field
cutouts at 8192 per call (index ranges per rank, no collective).  With N > 1 the line carries `multi_rank`: the
communicator size RCCL itself reports on every rank (ncclCommCount), each rank's device (PCI bus id), and the
communication of a step - time inside the all-reduces against the part the main stream waited for.

One process per GPU.  The compute path is the HIP engine (libdebvader_hip.so through ctypes) with RCCL
gradient all-reduce.  Rank 0's RCCL id, the barriers around the timed region and the max-over-ranks of the
elapsed time travel over debvader_amd.parallel.HostGroup (a TCP star on MASTER_ADDR): no torch in a GPU process.
Inputs are resident in HBM before the timed region (dv_data_upload), synthetic, random-init weights.

The ONE JSON line also carries (rank 0, N = 1 only; --no-secondary / --no-cpu-baseline / --no-roofline switch them off):
  roofline      per-kernel-family HIP-event timing of the same K steps with the engine's streams serialised
  cpu_baseline  the same train step restated with torch-CPU ops (tests/torch_ref.py, oneDNN, all host cores), and the
                numpy oracle as a second entry
  secondary     the other BASELINE configs, each measured here: bf16 train step (configs[2], HBM roofline), stage-2
                train step (decoder frozen, train.py:175-183), 128 x 128 x 6 / 6-level net (configs[3]), deblend()
                inference (configs[4]: cutouts of a tiled scene, 8192 per call)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0           # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
# minimum materialised activations 836 438 elements per stamp, five passes per train step
ACT_ELEMS_PER_STAMP = 836438
TRAIN_PASSES = 5
PARAM_STEP_BYTES = 5 * 33.27e6  # weights / gradients / Adam slots per step (fp32 in both engines)


_T0 = time.perf_counter()


def _progress(msg):
    """stderr only (stdout carries the one JSON line): where a long run is"""
    print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def _cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota (a GPU box shows 256
    cores in its mask but grants a share of 16 per GPU: 256 threads on that share ran the torch step 500x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, q // int(fh.read().split()[0])))
            break
        except Exception:
            continue
    return min(n, int(os.environ.get("DV_BENCH_MAX_THREADS", "16")))


def cpu_baseline_torch(batch: int, steps: int):
    """The train step (forward + ELBO + autograd backward + Adam) restated with torch CPU ops - conv2d /
    conv_transpose2d on oneDNN, float32, all host cores.  Not the reference (TensorFlow is not installable here), but
    the same arithmetic on the kind of CPU backend the reference's Keras fit() uses (train.py:27-37)."""
    import torch

    from oracle import vae_oracle as vo
    from tests import torch_ref as tr
    from debvader_amd.data import synthetic_stamps

    cores = _cores()
    torch.set_num_threads(cores)
    arch = vo.Arch()
    p = {k: torch.tensor(v, dtype=torch.float32, requires_grad="moving" not in k)
         for k, v in vo.init_params(arch, 0, dtype=np.float32).items()}
    x, y = synthetic_stamps(batch, seed=7)
    x, y = torch.tensor(x), torch.tensor(y)
    eps = torch.randn(batch, arch.latent_dim, generator=torch.Generator().manual_seed(0))
    opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=1e-4, eps=1e-7)

    def step():
        opt.zero_grad(set_to_none=True)
        out = tr.net_loss(arch, p, x, y, eps, training=True)
        out["loss"].backward()
        opt.step()

    step()                                            # warm-up (thread pool, oneDNN primitive cache)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "stamps/s", "cores": cores, "kind": "port",
            "impl": "torch-cpu restatement of the train step (tests/torch_ref.py); the reference's TensorFlow is not installable here",
            "sample": f"{steps} train steps of batch {batch} (torch {torch.__version__} CPU / oneDNN float32 conv2d + "
                      f"conv_transpose2d, autograd, Adam; tests/torch_ref.py), same model/config"}


def cpu_baseline_numpy(batch: int, steps: int):
    """The oracle's train step (numpy float32, BLAS threads) on a bounded sample."""
    from oracle import vae_oracle as vo
    from debvader_amd.data import synthetic_stamps

    cores = _cores()
    threads = min(cores, 32)         # numpy's BLAS does not scale past a few tens of threads on these GEMM sizes
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=threads)
    except Exception:                # pragma: no cover
        limiter, threads = None, cores
    arch = vo.Arch()
    p = vo.init_params(arch, 0, dtype=np.float32)
    x, y = synthetic_stamps(batch, seed=7)
    eps = np.random.default_rng(0).normal(size=(batch, arch.latent_dim)).astype(np.float32)
    st = vo.AdamState()
    vo.train_step(arch, p, st, x, y, eps)            # warm-up (BLAS thread pool, page faults)
    t0 = time.perf_counter()
    for _ in range(steps):
        vo.train_step(arch, p, st, x, y, eps)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": batch * steps / dt, "unit": "stamps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps of batch {batch} (numpy float32 oracle, BLAS multi-threaded), same model/config"}


def _git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def _pmc_traffic():
    """HBM bytes per step and kernel family from the committed rocprofv3 PMC passes of this round (FETCH_SIZE doubled as
    the micro-architecture guide prescribes for gfx950; tools/pmc_traffic.py spells out the collection)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                return json.load(fh), "profiles/" + name
        except Exception:
            continue
    return None, None


def _timed_steps(eng, ctx, B, steps, warmup, seed=100, Bg=None):
    if warmup > 0:
        eng.train_steps(0, 0, B, warmup, global_batch=Bg, seed=seed + 1)
    ctx.sync()
    t0 = time.perf_counter()
    scal = eng.train_steps(0, 0, B, steps, global_batch=Bg, seed=seed)
    ctx.sync()
    return time.perf_counter() - t0, scal


def _family_table(eng, B, steps, peak_tflops, seed):
    """Per-kernel-family rows from a serialised-stream pass over `steps` train steps."""
    eng.prof_reset()
    eng.prof_enable(True)
    eng.train_steps(0, 0, B, steps, seed=seed)
    eng.prof_enable(False)
    fams = eng.prof_families()
    rows = []
    for f in fams:
        ms = f["ms"] / steps
        tf = f["flops"] / steps / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        rows.append({"kernel": f["name"], "launches_per_step": f["launches"] / steps, "flops_per_step": f["flops"] / steps,
                     "executed_flops_per_step": f.get("executed_flops", f["flops"]) / steps,
                     "algorithmic_bytes_per_step": (f.get("algorithmic_bytes") or 0.0) / steps or None,
                     "ms_per_step": ms, "avg_us": ms * 1e3 / max(1.0, f["launches"] / steps), "tflops": tf,
                     "frac": tf / peak_tflops})
    classes = {k: eng.prof_read(i)[1] / steps for i, k in enumerate(("conv", "wgrad", "other"))}
    return rows, classes


def secondary_entries(E, ctx, synthetic_stamps, quick: bool):
    out = {}
    enc_macs, dec_macs = E.arch_macs(E.make_config())
    fwd_flops = 2.0 * (enc_macs + dec_macs + 32 * 33 // 2)
    # ---- BASELINE configs[2]: the same model with bf16 storage / bf16 MFMA operands (per-GPU batch 256) ----
    B = 256
    from debvader_amd.data import bench_stamps

    x, y, _ = bench_stamps(4 * B, seed=1000)
    steps = 20 if quick else 60
    _progress("secondary: bf16 train step")
    eng = E.Engine(E.make_config(max_batch=B, dtype=1), ctx)
    eng.init(seed=0)
    eng.upload(0, x, y)
    eng.optimizer_reset(1e-4)
    dt, scal = _timed_steps(eng, ctx, B, steps, 10)
    rows, classes = _family_table(eng, B, 10, BF16_MFMA_PEAK_TFLOPS, 300)
    eng.close()
    ms = dt / steps * 1e3
    alg_bytes = ACT_ELEMS_PER_STAMP * 2 * TRAIN_PASSES * B + PARAM_STEP_BYTES
    pmc, src = _pmc_traffic()
    out["bf16_train"] = {
        "workload": "BASELINE configs[2] per GPU: same model, bf16 storage + bf16 MFMA operands (conv stacks and, since r06, "
                    "the two large Dense layers of the trunk), fp32 accumulation / master weights / sampler / head, batch 256",
        "value": B * steps / dt, "unit": "stamps/s", "ms_per_step": ms, "dtype": "bf16", "last_loss": scal["loss"],
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_step": alg_bytes,
                     "traffic": (pmc or {}).get("bf16_per_step_bytes", {}).get("total"),
                     "traffic_source": src if (pmc or {}).get("bf16_per_step_bytes") else None,
                     "mfma_whole_step_tflops": 3.0 * fwd_flops * B / (ms * 1e-3) / 1e12,
                     "mode": "value: streams overlapped; kernels / classes: streams serialised",
                     "kernels": rows, "classes_ms_per_step": classes},
    }
    # ---- stage 2 of train_deblender: decoder frozen, fresh Adam (train.py:175-183); 1.509 GFLOP per stamp ----
    _progress("secondary: stage-2 train step")
    eng = E.Engine(E.make_config(max_batch=B), ctx)
    eng.init(seed=0)
    eng.upload(0, x, y)
    eng.set_trainable(True, False)
    eng.optimizer_reset(1e-4)
    dt, scal = _timed_steps(eng, ctx, B, steps, 10)
    ms = dt / steps * 1e3
    s2_flops = (2.0 * fwd_flops + 2.0 * enc_macs) * B          # fwd + all data gradients + encoder weight gradients
    out["stage2_train"] = {
        "workload": "stage 2 of train_deblender (train.py:175-183): decoder frozen, encoder trained, fp32, batch 256",
        "value": B * steps / dt, "unit": "stamps/s", "ms_per_step": ms, "dtype": "f32", "last_loss": scal["loss"],
        "whole_step_tflops": s2_flops / (ms * 1e-3) / 1e12,
        "frac_of_fp32_mfma_peak": s2_flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    eng.close()
    # ---- BASELINE configs[4]: deblend() over cutouts of a field_img_2.npy-style scene, 8192 stamps per call ----
    from tools.field_cutouts import run as cutouts_run

    # the same work as ONE engine call: gather + float32 cast on the GPU, results consumed chunk by chunk from the pinned
    # transfer ring (deblend_field_cutouts(on_chunk=...)); `python tools/field_cutouts.py --stream` runs the full million
    _progress("secondary: deblend_field_cutouts, streaming (fp32, then bf16)")
    out["deblend_field_cutouts_stream"] = cutouts_run(ctx, n_cutouts=65536 if quick else 262144, chunk=8192, dtype=0, stream=True)
    out["deblend_field_cutouts_stream_bf16"] = cutouts_run(ctx, n_cutouts=65536 if quick else 262144, chunk=8192, dtype=1,
                                                           stream=True)
    # ... and with the consumer that follows in the reference (get_predicted_field / get_residual_field) on the GPU as well:
    # no stamp crosses the host link (DeblendField.deblend_field(on_device=True), dv_infer_cutouts_composite)
    _progress("secondary: deblend_field on the device (fp32, then bf16)")
    # (the full million of configs[4] unless --quick: the call has a fixed cost of three 206 MB float64 fields coming back)
    out["deblend_field_on_device"] = cutouts_run(ctx, n_cutouts=131072 if quick else 1000000, chunk=8192, dtype=0, on_device=True)
    out["deblend_field_on_device_bf16"] = cutouts_run(ctx, n_cutouts=131072 if quick else 1000000, chunk=8192, dtype=1,
                                                      on_device=True)
    _progress("secondary: deblend over field cutouts (fp32, then bf16)")
    # the reference's own call sequence (DeblendField.deblend_field with its defaults): one engine call per deblend_field
    from tools.field_cutouts import run_drop_in

    out["deblend_cutouts"] = run_drop_in(ctx, n_cutouts=32768 if quick else 131072, chunk=8192, dtype=0,
                                         per_call=16384 if quick else 32768)
    out["deblend_cutouts_bf16"] = run_drop_in(ctx, n_cutouts=32768 if quick else 131072, chunk=8192, dtype=1,
                                              per_call=16384 if quick else 32768)
    # ---- BASELINE configs[3]: 128 x 128 x 6 stamps, six levels (per-GPU share of the global batch 512: 64) ----
    _progress("secondary: 128-pixel architecture")
    B3 = 64
    cfg3 = E.make_config((128, 128, 6), 32, (32, 64, 128, 256, 512, 512), (3,) * 6, max_batch=B3)
    e3, d3 = E.arch_macs(cfg3)
    rng = np.random.default_rng(5)
    x3 = rng.normal(0, 0.3, size=(2 * B3, 128, 128, 6)).astype(np.float32)
    y3 = np.abs(x3) * 0.5
    eng = E.Engine(cfg3, ctx)
    eng.init(seed=0)
    eng.upload(0, x3, y3)
    eng.optimizer_reset(1e-4)
    st3 = 6 if quick else 20
    dt, scal = _timed_steps(eng, ctx, B3, st3, 3)
    ms = dt / st3 * 1e3
    fl3 = 3.0 * 2.0 * (e3 + d3) * B3
    out["arch128_train"] = {
        "workload": "BASELINE configs[3] per GPU: 128x128x6 stamps, 6 levels, filters [32,64,128,256,512,512], fp32, batch 64",
        "value": B3 * st3 / dt, "unit": "stamps/s", "ms_per_step": ms, "dtype": "f32", "last_loss": scal["loss"],
        "whole_step_tflops": fl3 / (ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": fl3 / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    eng.close()
    return out


ARCH128 = dict(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6)
CONFIGS = {
    1: dict(dtype=0, batch=256, arch={}, label="BASELINE configs[1]: 6-band 59x59 stamps, batch=256 per GPU, latent_dim=32, "
                                               "filters [32,64,128,256], fp32, stage-1 VAE train step"),
    2: dict(dtype=1, batch=256, arch={}, label="BASELINE configs[2]: same model and step, bf16 storage + bf16 MFMA operands, fp32 "
                                               "accumulation / master weights / head, batch=256 per GPU"),
    3: dict(dtype=0, batch=64, arch=ARCH128, label="BASELINE configs[3]: 128x128x6 stamps, 6 levels, filters "
                                                   "[32,64,128,256,512,512], fp32, batch=64 per GPU (512 on 8 GPUs)"),
    4: dict(dtype=0, batch=8192, arch={}, label="BASELINE configs[4]: deblend() inference over cutouts of a field_img_2.npy-style "
                                                "scene, 8192 per network call, index ranges sharded over the GPUs, no collective"),
}


def _roofline_f32(rows, classes, pmc, src, B, dt, steps, train_flops):
    """The roofline object of an fp32 train configuration (MFMA-bound), from the per-family rows of the serialised pass.
    `frac` is PHYSICAL: the FLOPs the matrix pipe executes for the dominant kernel's launches (Winograd-domain multiplies,
    block / column-tile padding included) over their duration over the dense fp32 MFMA peak - what SQ_VALU_MFMA_BUSY sees.
    The direct-convolution rate SURVEY 8(d) prices the layers with is carried beside it as algorithmic_*."""
    per_kernel = (pmc or {}).get("per_kernel_bytes", {})
    for r in rows:
        ms = r["ms_per_step"]
        r["algorithmic_tflops"] = r["tflops"]
        r["algorithmic_frac"] = r["tflops"] / FP32_MFMA_PEAK_TFLOPS
        r["executed_tflops"] = r["executed_flops_per_step"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r["executed_frac"] = r["executed_tflops"] / FP32_MFMA_PEAK_TFLOPS
        r["frac"] = r["executed_frac"]
        if r["kernel"].startswith("wino_"):
            r["note"] = ("Winograd: algorithmic_* price the launch as a direct convolution (SURVEY 8(d)); executed_* count the "
                         "16 multiplies per 2x2 tile and channel pair the matrix pipe really does, padding included")
    dom = max(rows, key=lambda r: r["ms_per_step"]) if rows else None
    conv_rows = [r for r in rows if r["kernel"].startswith("gconv") or r["kernel"].startswith("wino_conv")]
    conv_ms = sum(r["ms_per_step"] for r in conv_rows)
    conv_fl = sum(r["flops_per_step"] for r in conv_rows)
    conv_ex = sum(r["executed_flops_per_step"] for r in conv_rows)
    traffic = None
    if dom is not None:
        # HBM bytes of the same launches (one step's launches of the dominant kernel), from the committed PMC passes
        # (a row may stand for several rocprof kernel names, "a / b": their bytes add up)
        parts = [per_kernel.get(n.strip()) for n in dom["kernel"].split("/")]
        if any(parts):
            traffic = sum((q or {}).get("hbm_bytes", 0.0) for q in parts)
    alg_bytes = dom["algorithmic_bytes_per_step"] if dom and dom.get("algorithmic_bytes_per_step") else None
    whole = train_flops * B / (dt / steps) / 1e12
    return {
        "bound": "mfma", "kernel": dom["kernel"] if dom else None,
        "achieved": dom["executed_tflops"] if dom else None, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": dom["executed_frac"] if dom else None,
        "frac_definition": "executed matrix FLOPs of the dominant kernel's launches (tile / block padding included) / their "
                           "duration / 157.3 TFLOP/s; comparable with SQ_VALU_MFMA_BUSY of profiles/*_pmc_f32_mfma.json",
        "algorithmic_tflops": dom["algorithmic_tflops"] if dom else None,
        "algorithmic_frac": dom["algorithmic_frac"] if dom else None,
        "algorithmic_bytes": alg_bytes, "traffic": traffic,
        "traffic_over_algorithmic": (traffic / alg_bytes) if traffic and alg_bytes else None,
        "traffic_source": {"file": src, "commit": _git_head(),
                           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed; not "
                                   "re-measured in this run"} if src and traffic else None,
        "mode": "streams serialised (the per-kernel rows); `value` is measured with the streams overlapped",
        "launch": (f"one training step's launches of {dom['kernel']} ({dom['launches_per_step']:.0f} launches, "
                   f"batch {B})") if dom else None,
        "kernels": rows,
        "conv_family": {"kernels": "gconv* + wino_conv (forward and data-gradient launches)", "ms_per_step": conv_ms,
                        "flops_per_step": conv_fl,
                        "algorithmic_tflops": conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                        "executed_frac": conv_ex / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS if conv_ms > 0 else 0.0},
        "classes_ms_per_step": classes,
        "whole_step_tflops": whole, "whole_step_frac": whole / FP32_MFMA_PEAK_TFLOPS,
        "whole_step_note": "algorithmic train FLOPs (3 x forward, SURVEY 8(d)) of the overlapped step over the fp32 MFMA peak",
    }


def _same_gpu_rehearsal():
    """DV_DEBUG_SAME_GPU is honoured by the development library only (parallel._rehearsal)"""
    from debvader_amd import parallel
    return parallel._rehearsal("DV_DEBUG_SAME_GPU")


def _multi_rank_block(ctx, group, eng, B, Bg, steps, world):
    """What makes an N > 1 line checkable: the size RCCL itself reports for the communicator on every rank, the device
    each rank drives, and the communication of a step - time inside the collectives on the comm stream against the part
    of it the main stream actually waited for (a separate pass with timed events: they perturb the step a little)."""
    info = ctx.comm_info() if hasattr(ctx, "comm_info") else {}
    prof = None
    if eng is not None and hasattr(ctx, "comm_prof"):
        ctx.comm_prof(True)
        eng.train_steps(0, 0, B, steps, global_batch=Bg, seed=300)
        prof = ctx.comm_prof_read()
        ctx.comm_prof(False)
    mine = {"info": info, "prof": prof}
    allr = group.gather_object(mine, dst=0) if group is not None else [mine]
    if allr is None:
        return None
    infos = [a["info"] for a in allr]
    profs = [a["prof"] for a in allr if a["prof"]]
    blk = {
        "world": world,
        "rccl_ranks": [i.get("comm_ranks") for i in infos],          # ncclCommCount on every rank: must all equal world
        "rccl_rank_ids": [i.get("comm_rank") for i in infos],
        "devices": [{"rank": r, "hip_device": i.get("device"), "pci_bus_id": i.get("bus_id")} for r, i in enumerate(infos)],
        "distinct_devices": len({i.get("bus_id") for i in infos}),
        "rehearsal": any(i.get("rehearsal") for i in infos) or _same_gpu_rehearsal(),
    }
    if profs:
        blk.update({
            "collectives_per_step": profs[0]["collectives"] / steps,
            "comm_ms_per_step": max(p["comm_ms"] for p in profs) / steps,
            "exposed_comm_ms_per_step": max(p["exposed_ms"] for p in profs) / steps,
            "comm_note": "max over ranks; comm = time inside all-reduces on the comm stream (3 gradient buckets, BN sums, "
                         "loss sums), exposed = what the main stream waited for them; separate pass with timed events",
        })
    blk["verified"] = (not blk["rehearsal"]) and all(n == world for n in blk["rccl_ranks"]) and blk["distinct_devices"] == world
    return blk


def run_inference_config(args, E, ctx, group, rank, world):
    """BASELINE configs[4]: a "step" is one 8192-cutout network call per GPU; rank r takes the contiguous index range
    parallel.shard_range(N, r, world) of the cutout list (no collective on the data path).  What is timed is
    the reference's own chain for a field (deblend/field_deblender.py:219-383 and :46-189): cut the stamps out of the
    field, run the network on them, composite the predicted mean / stddev / residual fields - here as ONE engine call per
    rank with everything on the GPU (DeblendField.deblend_field(on_device=True)); each rank returns its partial fields.
    The form that ships every stamp's mean and stddev to the host instead (deblend_field_cutouts(on_chunk=...), 167 KB
    per stamp over the host link) is measured beside it as `stamps_to_host`."""
    from tools.field_cutouts import synthetic_field
    from debvader_amd.parallel import shard_range

    chunk = args.batch or CONFIGS[4]["batch"]
    dtype = 1 if args.dtype == "bf16" else 0
    scene = np.ascontiguousarray(np.tile(synthetic_field(), (8, 8, 1)))
    F, cs = scene.shape[0], 59
    n_total = chunk * world * (args.steps + max(args.warmup, 2))
    starts = np.random.default_rng(0).integers(0, F - cs + 1, size=(n_total, 2)).astype(np.int32)
    eng = E.Engine(E.make_config(max_batch=chunk, dtype=dtype), ctx)
    eng.init(seed=0)

    def barrier():
        ctx.sync()
        if group is not None:
            group.barrier()

    n_timed = chunk * world * args.steps
    off = n_total - n_timed
    wlo, whi = shard_range(off, rank, world)
    lo, hi = shard_range(n_timed, rank, world)
    eng.infer_cutouts_composite(scene, starts[wlo:whi], starts[wlo:whi], seed=1)                    # warm-up
    barrier()
    t0 = time.perf_counter()
    out = eng.infer_cutouts_composite(scene, starts[off + lo:off + hi], starts[off + lo:off + hi], seed=2)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if group is not None:
        dt = group.max(dt)
    # the same cutouts with every stamp's mean and stddev shipped to the host (PCIe-inclusive)
    state = {"n": 0}

    def consume(first, mean, std):
        state["n"] += mean.shape[0]

    eng.infer_cutouts_stream(scene, starts[wlo:whi], lambda *a: None, seed=1)
    barrier()
    t1 = time.perf_counter()
    eng.infer_cutouts_stream(scene, starts[off + lo:off + hi], consume, seed=2)
    ctx.sync()
    barrier()
    dt_host = time.perf_counter() - t1
    if group is not None:
        dt_host = group.max(dt_host)
    rows = None
    if rank == 0 and not args.no_roofline and hasattr(eng, "prof_enable"):
        # per-kernel-family pass over two chunks (HIP events on the launch streams), as for the train configurations
        eng.prof_reset()
        eng.prof_enable(True)
        eng.infer_cutouts_composite(scene, starts[:2 * chunk], starts[:2 * chunk], seed=3)
        eng.prof_enable(False)
        rows = []
        for f in eng.prof_families():
            ms = f["ms"] / 2
            rows.append({"kernel": f["name"], "launches_per_step": f["launches"] / 2, "flops_per_step": f["flops"] / 2,
                         "executed_flops_per_step": f["executed_flops"] / 2,
                         "algorithmic_bytes_per_step": (f["algorithmic_bytes"] / 2) or None, "ms_per_step": ms,
                         "avg_us": ms * 1e3 / max(1.0, f["launches"] / 2),
                         "tflops": f["flops"] / 2 / (ms * 1e-3) / 1e12 if ms > 0 else 0.0})
    eng.close()
    mr = _multi_rank_block(ctx, group, None, 0, 0, 1, world) if world > 1 else None     # who ran where (no timing pass)
    if rank != 0:
        return None
    enc_macs, dec_macs = E.arch_macs(E.make_config())
    fwd_flops = 2.0 * (enc_macs + dec_macs + 32 * 33 // 2)
    val = n_timed / dt
    peak = BF16_MFMA_PEAK_TFLOPS if dtype else FP32_MFMA_PEAK_TFLOPS
    whole = val * fwd_flops / world / 1e12
    roofline = {"bound": "mfma", "kernel": None, "achieved": whole, "peak": peak, "unit": "TFLOP/s", "frac": whole / peak,
                "traffic": None, "note": "whole forward, algorithmic FLOPs per GPU (658.7 MFLOP per stamp) over the dense MFMA peak"}
    if rows:
        dom = max(rows, key=lambda r: r["ms_per_step"])
        for r in rows:
            r["algorithmic_tflops"] = r["tflops"]
            r["executed_tflops"] = r["executed_flops_per_step"] / (r["ms_per_step"] * 1e-3) / 1e12 if r["ms_per_step"] > 0 else 0.0
            r["executed_frac"] = r["executed_tflops"] / peak
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["executed_tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": dom["executed_frac"],
                    "frac_definition": "executed matrix FLOPs of the dominant kernel's launches of one 8192-cutout call / their "
                                       "duration / the dense MFMA peak of the dtype",
                    "algorithmic_tflops": dom["algorithmic_tflops"], "algorithmic_frac": dom["algorithmic_tflops"] / peak,
                    "algorithmic_bytes": dom["algorithmic_bytes_per_step"], "traffic": None,
                    "whole_step_tflops": whole, "whole_step_frac": whole / peak, "kernels": rows}
    return {
        "metric": "galaxy stamps/sec (deblend_field inference over field cutouts, fields composited on the GPU) 59x59x6",
        "value": val, "unit": "stamps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if dtype else "f32", "data": "synthetic",
        "config": {"workload": CONFIGS[4]["label"], "global_batch": chunk * world, "per_gpu_batch": chunk,
                   "parallelism": f"shard{world} (no collective)"},
        "includes": "the field is in HBM when the timed region starts (uploaded by the call: 206 MB once per rank); cutout gather + "
                    "float32 cast, forward, mean / stddev / residual fields composited on the GPU; D2H of the three fields",
        "stamps_to_host": {"value": n_timed / dt_host, "unit": "stamps/s", "d2h_gbs": n_timed / dt_host * 2 * cs * cs * 6 * 4 / 1e9,
                           "note": "the same cutouts with mean and stddev of every stamp copied to the host (167 KB per stamp: "
                                   "PCIe-inclusive, bound by the box's host link)"},
        "roofline": roofline, "cpu_baseline": None, "checksum": float(out["mean_field"][::37, ::41, 2].sum()),
        **({"multi_rank": mr} if mr else {}),
    }


_json_out = sys.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, choices=(1, 2, 3, 4), default=None,
                    help="BASELINE configs[k]: 1 fp32 59 px batch 256 (the headline, default), 2 the same step on the bf16 "
                         "engine, 3 128x128x6 / six levels at 64 stamps per GPU (512 on 8), 4 sharded deblend() inference at "
                         "8192 cutouts per call (no collective)")
    ap.add_argument("--batch", type=int, default=None, help="stamps per GPU per step (default: the configuration's)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None,
                    help="bf16 without --config selects configs[2]; with --config 4 the bf16 engine's inference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--quick", action="store_true", help="shorter secondary measurements")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_torch(batch=256, steps=5)), flush=True)
        return
    # stdout carries ONE JSON line and nothing else (the driver parses it): everything the measured code prints on the way -
    # e.g. create_model_vae's "in cropping", as the reference prints it (model.py:142) - goes to stderr
    global _json_out
    _json_out = sys.stdout
    sys.stdout = sys.stderr
    if args.config is None:
        args.config = 2 if args.dtype == "bf16" else 1
    conf = CONFIGS[args.config]
    if args.dtype is None:
        args.dtype = "bf16" if conf["dtype"] else "f32"

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    from debvader_amd import engine as E
    from debvader_amd.data import synthetic_stamps
    from debvader_amd import parallel

    stub = os.environ.get("DV_BENCH_STUB_ENGINE")        # CPU test of the rank plumbing (tests/test_bench_plumbing.py)
    if stub:
        import importlib

        E = importlib.import_module(stub)
        parallel.E = E
        args.no_roofline = args.no_secondary = args.no_cpu_baseline = True

    ctx = parallel.make_context(rank, world, local_rank)     # world > 1: the ranks meet in a parallel.HostGroup
    group = ctx.group

    if args.config == 4:
        line = run_inference_config(args, E, ctx, group, rank, world)
        if rank == 0:
            print(json.dumps(line), file=_json_out, flush=True)
        if group is not None:
            group.barrier()
        ctx.close()
        if group is not None:
            group.close()
        return

    B = args.batch or conf["batch"]
    bf16 = args.dtype == "bf16"
    cfg = E.make_config(max_batch=B, dtype=1 if bf16 else 0, **conf["arch"])
    eng = E.Engine(cfg, ctx)
    eng.init(seed=0)                                     # same weights on every rank
    pool = 4 * B                                         # per-rank shard of the synthetic set, resident in HBM
    if conf["arch"]:
        H, C = cfg.height, cfg.bands
        rng = np.random.default_rng(5 + rank)
        x = rng.normal(0, 0.3, size=(2 * B, H, H, C)).astype(np.float32)
        y = np.abs(x) * 0.5
        pool = 2 * B
        data_desc = "synthetic (Gaussian noise stamps of the 128-px geometry); random-init weights"
    else:
        # the real DC2 sample stamps tiled + the config-1 generator to fill the resident pool
        from debvader_amd.data import bench_stamps

        x, y, data_desc = bench_stamps(pool, seed=1000 + rank)
    eng.upload(0, x, y)
    eng.optimizer_reset(1e-4)
    Bg = B * world

    def barrier():
        ctx.sync()
        if group is not None:
            group.barrier()

    if args.warmup > 0:
        eng.train_steps(0, 0, B, args.warmup, global_batch=Bg, seed=1)
    barrier()
    t0 = time.perf_counter()
    scal = eng.train_steps(0, 0, B, args.steps, global_batch=Bg, seed=100)   # returns after the stream has drained
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if group is not None:
        dt = group.max(dt)

    enc_macs, dec_macs = E.arch_macs(cfg)
    fwd_flops = 2.0 * (enc_macs + dec_macs + cfg.latent_dim * (cfg.latent_dim + 1) // 2)
    train_flops = 3.0 * fwd_flops                         # fwd + dgrad + wgrad

    _progress(f"headline: {Bg * args.steps / dt:.0f} stamps/s")
    multi = None
    if world > 1:
        multi = _multi_rank_block(ctx, group, eng, B, Bg, min(args.steps, 20), world)
    roofline = None
    if world > 1 and rank == 0:
        whole = train_flops * B / (dt / args.steps) / 1e12      # per GPU
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        roofline = {"bound": "hbm" if bf16 else "mfma", "kernel": None, "achieved": whole, "peak": peak, "unit": "TFLOP/s",
                    "frac": whole / peak, "traffic": None, "whole_step_tflops": whole, "whole_step_frac": whole / peak,
                    "note": "N > 1: whole-step algorithmic train FLOPs PER GPU over the dense MFMA peak of the dtype (the "
                            "per-kernel rows need the serialised single-GPU pass: run with --gpus 1)"}
    if not args.no_roofline and world == 1:
        # HIP-event timing per kernel family on the stream each launch is queued on, over K steps of the same workload,
        # with the engine's streams SERIALISED (a separate pass: the records would perturb `value`, and overlapped
        # kernels cannot be told apart by events)
        rows, classes = _family_table(eng, B, args.steps, BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS, 200)
        pmc, src = _pmc_traffic()
        if conf["arch"]:
            pmc, src = None, None                       # the committed PMC passes are of the 59-px step
        if bf16:
            dom = max(rows, key=lambda r: r["ms_per_step"]) if rows else None
            alg_bytes = ACT_ELEMS_PER_STAMP * 2 * TRAIN_PASSES * B + PARAM_STEP_BYTES
            ms_step = dt / args.steps * 1e3
            traffic = (pmc or {}).get("bf16_per_step_bytes", {}).get("total")
            roofline = {
                "bound": "hbm", "kernel": dom["kernel"] if dom else None,
                "achieved": alg_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": alg_bytes, "traffic": traffic,
                "traffic_over_algorithmic": traffic / alg_bytes if traffic else None,
                "traffic_source": src if traffic else None,
                "launch": "one training step per GPU (all launches), algorithmic bytes 8.36 MB per stamp + 5 x 33.3 MB",
                "mode": "streams serialised (the per-kernel rows); `value` is measured with the streams overlapped",
                "kernels": rows, "classes_ms_per_step": classes,
                "mfma_whole_step_tflops": train_flops * B / (dt / args.steps) / 1e12,
            }
        else:
            roofline = _roofline_f32(rows, classes, pmc, src, B, dt, args.steps, train_flops)
    last_loss = scal["loss"]
    eng.close()

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and args.config == 1:
        secondary = secondary_entries(E, ctx, synthetic_stamps, args.quick)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config in (1, 2):
        # in a child process: torch brings its own copy of the HIP runtime, and a process that has loaded both it and
        # /opt/rocm's (libdebvader_hip.so) aborts in the static destructors at exit ("free(): invalid pointer")
        _progress("cpu baseline: torch-CPU restatement (child process)")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], capture_output=True,
                               text=True, timeout=900)
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                    # pragma: no cover
            cpu = {"value": None, "unit": "stamps/s", "cores": _cores(), "kind": "port",
                   "sample": f"failed: {e!r}"}
        _progress("cpu baseline: numpy oracle")
        cpu["also"] = cpu_baseline_numpy(batch=64, steps=3)
        _progress("done")

    if rank == 0:
        value = Bg * args.steps / dt
        rehearsal = bool(multi and multi.get("rehearsal"))
        stamp = f"{cfg.height}x{cfg.width}x{cfg.bands}"
        line = {
            "metric": f"galaxy stamps/sec (train fwd+bwd+Adam) {stamp}",
            # a REHEARSAL (every rank on one GPU, one-rank communicators: DV_DEBUG_SAME_GPU / DV_DEBUG_FAKE_PEERS) exercises
            # the launch line and the step structure; its throughput means nothing and is not reported as a value
            "value": None if rehearsal else value, "unit": "stamps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": data_desc,
            "config": {"workload": conf["label"], "global_batch": Bg, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "last_loss": last_loss,
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        if multi is not None:
            line["multi_rank"] = multi
        if rehearsal:
            line["rehearsal"] = True
            line["rehearsal_stamps_per_s"] = value
        print(json.dumps(line), file=_json_out, flush=True)
    if group is not None:
        group.barrier()          # nobody tears its communicator down while another rank is still inside a collective
    ctx.close()
    if group is not None:
        group.close()


if __name__ == "__main__":
    main()
