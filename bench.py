#!/usr/bin/env python3
"""Headline benchmark: galaxy stamps/sec of the conv-VAE training step (fwd + ELBO + bwd + Adam),
59x59x6 stamps, batch 256 per GPU, latent 32, fp32 (BASELINE.json configs[1]); weak scaling over N GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  The compute path is the HIP engine (libdebvader_hip.so through ctypes) with RCCL
gradient all-reduce; torch.distributed (gloo) is used only to hand rank 0's RCCL id to the other ranks,
for the barriers around the timed region and for the max-over-ranks of the elapsed time.
Inputs are resident in HBM before the timed region (dv_data_upload), synthetic, random-init weights.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak


def cpu_baseline(batch: int, steps: int):
    """The oracle's train step (numpy float32, BLAS threads = all host cores) on a bounded sample."""
    from oracle import vae_oracle as vo
    from debvader_amd.data import synthetic_stamps

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
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
    return {
        "value": batch * steps / dt,
        "unit": "stamps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{steps} train steps of batch {batch} (numpy float32 oracle, BLAS multi-threaded), same model/config",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="stamps per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

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

    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("gloo")
    ctx = parallel.make_context(rank, world, local_rank, dist)

    B = args.batch
    cfg = E.make_config(max_batch=B)
    eng = E.Engine(cfg, ctx)
    eng.init(seed=0)                                     # same weights on every rank
    pool = 4 * B                                         # per-rank shard of the synthetic set, resident in HBM
    x, y = synthetic_stamps(pool, seed=1000 + rank)
    eng.upload(0, x, y)
    eng.optimizer_reset(1e-4)
    Bg = B * world

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    if args.warmup > 0:
        eng.train_steps(0, 0, B, args.warmup, global_batch=Bg, seed=1)
    barrier()
    t0 = time.perf_counter()
    scal = eng.train_steps(0, 0, B, args.steps, global_batch=Bg, seed=100)   # returns after the stream has drained
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    enc_macs, dec_macs = E.arch_macs(cfg)
    fwd_flops = 2.0 * (enc_macs + dec_macs + cfg.latent_dim * (cfg.latent_dim + 1) // 2)
    train_flops = 3.0 * fwd_flops                         # fwd + dgrad + wgrad

    roofline = None
    if not args.no_roofline:
        # per-kernel-class HIP-event timing on the engine's stream, over the same K steps (separate pass so the
        # event records do not perturb `value`)
        eng.prof_reset()
        eng.prof_enable(True)
        eng.train_steps(0, 0, B, args.steps, global_batch=Bg, seed=200)
        eng.prof_enable(False)
        n_g, ms_g = eng.prof_read(0)
        n_w, ms_w = eng.prof_read(1)
        n_o, ms_o = eng.prof_read(2)
        # algorithmic FLOPs per step: the gather-GEMM kernel family carries fwd + dgrad (2/3), wgrad 1/3
        cls = {
            "gconv_kernel": (2.0 * fwd_flops * B, ms_g / args.steps, n_g // args.steps),
            "wgrad_kernel": (1.0 * fwd_flops * B, ms_w / args.steps, n_w // args.steps),
        }
        traffic = None
        try:    # HBM bytes per step of the family from the committed PMC passes (profiles/r01_pmc_traffic_v13.json)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_v13.json")) as fh:
                pmc = json.load(fh)["per_step_bytes"]
            traffic = {"gconv_kernel": next(v for k, v in pmc.items() if k.startswith("gconv"))["hbm_bytes"],
                       "wgrad_kernel": next(v for k, v in pmc.items() if k.startswith("wgrad"))["hbm_bytes"]}
        except Exception:
            traffic = None
        dom = max(cls, key=lambda k: cls[k][1])
        fl, ms, nl = cls[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        roofline = {
            "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
            "traffic": (traffic or {}).get(dom),
            "launch": f"one training step's launches of the {dom} family ({nl} launches, batch {B})",
            "flops_per_step": fl, "ms_per_step_in_kernel": ms,
            "classes_ms_per_step": {"gconv": ms_g / args.steps, "wgrad": ms_w / args.steps,
                                    "pointwise": ms_o / args.steps},
            "whole_step_tflops": train_flops * B / (dt / args.steps) / 1e12,
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batch=64, steps=3)

    if rank == 0:
        value = Bg * args.steps / dt
        line = {
            "metric": "galaxy stamps/sec (train fwd+bwd+Adam) 59x59x6",
            "value": value, "unit": "stamps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 6-band 59x59 stamps, batch=256 per GPU, latent_dim=32, "
                                   "filters [32,64,128,256], fp32, stage-1 VAE train step",
                       "global_batch": Bg, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "last_loss": scal["loss"],
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
