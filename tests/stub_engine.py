"""Stand-in for debvader_amd.engine in the CPU test of bench.py's rank plumbing (tests/test_bench_plumbing.py): no GPU,
no HIP.  It records what every rank did in $DV_STUB_LOG.<rank> so that the test can check the RCCL-id broadcast, the
barriers around the timed region and the MAX over ranks of the elapsed time."""
import json
import os
import sys
import time

import numpy as np

from debvader_amd import _lib
from debvader_amd.engine import arch_macs, make_config        # host-only queries of the real library

_LOG = os.environ.get("DV_STUB_LOG", "/tmp/dv_stub_log")


def _log(rank, **kw):
    with open(f"{_LOG}.{rank}", "a") as fh:
        fh.write(json.dumps(kw) + "\n")


def device_bus_id(device):
    """What the real engine reads with hipDeviceGetPCIBusId; $DV_STUB_BUS forces one id on every rank (two ranks on one GPU)."""
    return os.environ.get("DV_STUB_BUS", f"0000:{int(device):02x}:00.0")


def device_count():
    """GPUs visible to this process ($DV_STUB_VISIBLE; 0 = unknown, as on a box without a GPU)"""
    return int(os.environ.get("DV_STUB_VISIBLE", "0"))


class Context:
    def __init__(self, device=0, rank=0, world=1, unique_id=None):
        self.device, self.rank, self.world = device, rank, world
        if world > 1 and (unique_id is None or len(unique_id) != _lib.DV_UNIQUE_ID_BYTES):
            raise ValueError("world > 1 needs rank 0's unique id")
        _log(rank, event="ctx", world=world, device=device, uid=(unique_id or b"").hex())

    @staticmethod
    def unique_id() -> bytes:
        return bytes((7 * i + 3) % 256 for i in range(_lib.DV_UNIQUE_ID_BYTES))

    def sync(self):
        _log(self.rank, event="sync", t=time.time())

    def comm_info(self):
        return dict(comm_ranks=self.world if self.world > 1 else 0, comm_rank=self.rank, device=self.device,
                    bus_id=f"0000:{self.device:02x}:00.0", rehearsal=False, world=self.world, rank=self.rank)

    def comm_prof(self, on):
        self._prof = bool(on)

    def comm_prof_read(self):
        return dict(collectives=5 * getattr(self, "_steps", 0), comm_ms=0.4 * getattr(self, "_steps", 0),
                    waits=2 * getattr(self, "_steps", 0), exposed_ms=0.05 * (1 + self.rank) * getattr(self, "_steps", 0))

    def close(self):
        _log(self.rank, event="close", torch_loaded="torch" in sys.modules)


class Engine:
    def __init__(self, cfg, ctx=None):
        self.cfg, self.ctx = cfg, ctx
        self.max_batch = cfg.max_batch

    def init(self, seed=0):
        _log(self.ctx.rank, event="init", seed=seed)

    def upload(self, slot, x, y):
        return x.shape[0]

    def optimizer_reset(self, lr=1e-4, *a):
        pass

    def train_steps(self, slot, first, B, steps, global_batch=None, seed=0):
        # rank r is slower by 20 ms per step: the reported time must be the slowest rank's
        if getattr(self.ctx, "_prof", False):
            self.ctx._steps = steps
        time.sleep(steps * (0.005 + 0.02 * self.ctx.rank))
        _log(self.ctx.rank, event="train_steps", B=B, steps=steps, global_batch=global_batch, t=time.time(),
             torch_loaded="torch" in sys.modules)
        return {"loss": 1.0, "nll_mean": 1.0, "kl_reg": 0.0, "mse": 0.0}

    def close(self):
        pass
