"""Thin numpy-level wrapper over the C-ABI (one Context per process/GPU, one Engine per network).

This is the only module that touches ctypes pointers; the Keras-like objects in debvader_amd.model
sit on top of it.
"""
from __future__ import annotations

import atexit
import collections
import ctypes as C
import os
import sys
import threading
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import DvConfig, check, lib


def make_config(input_shape=(59, 59, 6), latent_dim=32, filters=(32, 64, 128, 256), kernels=(3, 3, 3, 3),
                max_batch=256, **overrides) -> DvConfig:
    cfg = DvConfig()
    check(lib.dv_config_default(C.byref(cfg)))
    if len(filters) != len(kernels):
        raise ValueError("filters and kernels must have the same length")
    if len(filters) > _lib.DV_MAX_LEVELS:
        raise ValueError(f"at most {_lib.DV_MAX_LEVELS} levels")
    cfg.height, cfg.width, cfg.bands = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
    cfg.latent_dim = int(latent_dim)
    cfg.n_levels = len(filters)
    for i in range(_lib.DV_MAX_LEVELS):
        cfg.filters[i] = int(filters[i]) if i < len(filters) else 0
        cfg.kernels[i] = int(kernels[i]) if i < len(kernels) else 0
    cfg.max_batch = int(max_batch)
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


def arch_specs(cfg: DvConfig) -> List[Tuple[str, Tuple[int, ...], bool]]:
    """(name, shape, trainable) of every tensor in TF-checkpoint order. Host only."""
    n = C.c_int32()
    check(lib.dv_arch_counts(C.byref(cfg), C.byref(n), None, None, None))
    out = []
    for i in range(n.value):
        name = C.create_string_buffer(128)
        shape = (C.c_int64 * 4)()
        nd, tr = C.c_int32(), C.c_int32()
        check(lib.dv_arch_describe(C.byref(cfg), i, name, 128, shape, C.byref(nd), C.byref(tr)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), bool(tr.value)))
    return out


def arch_counts(cfg: DvConfig) -> Dict[str, int]:
    n = C.c_int32()
    e, d, t = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.dv_arch_counts(C.byref(cfg), C.byref(n), C.byref(e), C.byref(d), C.byref(t)))
    return dict(tensors=n.value, encoder=e.value, decoder=d.value, trainable=t.value)


def arch_macs(cfg: DvConfig) -> Tuple[int, int]:
    e, d = C.c_int64(), C.c_int64()
    check(lib.dv_arch_macs(C.byref(cfg), C.byref(e), C.byref(d)))
    return e.value, d.value


def device_count() -> int:
    n = C.c_int32()
    check(lib.dv_device_count(C.byref(n)))
    return n.value


def device_bus_id(device: int) -> str:
    """PCI bus id of visible device `device` ("" when it cannot be had: no GPU, index not visible)."""
    buf = C.create_string_buffer(32)
    if lib.dv_device_bus_id(int(device), buf, 32) != 0:
        return ""
    return buf.value.decode()


def _fp(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32_rows(a, what: str) -> np.ndarray:
    """Integer (x, y) rows for the C-ABI's int32 arrays.  A plain astype would wrap 64-bit values into range silently -
    a start of 2**32 + 5 would fetch the cutout at 5 - so the range is checked before the cast."""
    a = np.asarray(a)
    if a.dtype.kind not in "iu":
        r = np.rint(np.asarray(a, dtype=np.float64))
        if a.size and not np.array_equal(r, np.asarray(a, dtype=np.float64)):
            raise ValueError(f"{what} must be integers")
        a = r
    if a.size and (a.min() < -2 ** 31 or a.max() > 2 ** 31 - 1):
        raise ValueError(f"{what} outside the int32 range")
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 2)


class _Lease:
    """Owner token of one hand-out of a pool block.  The array the caller receives is built on this object
    (__array_interface__), so numpy makes it the END of the `base` chain of that array and of everything derived from it -
    views, reshapes, the object columns of a recarray, a memoryview, np.frombuffer of one.  When the last of them dies the
    token dies, and its weakref.finalize puts the block back on the pool's return queue."""

    __slots__ = ("_mem", "__array_interface__", "__weakref__")

    def __init__(self, mem: np.ndarray, nbytes: int):
        self._mem = mem                                     # keeps the block alive while the lease is
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (mem.ctypes.data, False), "version": 3}


class _HostPool:
    """Recycles the host memory of large result arrays across calls.

    DeblendField.deblend_field returns 16 bytes per pixel and band and galaxy (float64 cutout, float32 mean and stddev): 11 GB
    per 32 768 galaxies.  Fresh np.empty arrays cost a page fault per 4 KiB while the engine's copy threads fill them and
    an munmap of the same size when the previous result is dropped - together more than the GPU work of the call
    (tools/probes/df_lines.py: 0.21 s engine call, 0.25 - 0.5 s freeing the previous recarray).

    Ownership (round 6; until then the pool looked at sys.getrefcount of its blocks, without a lock).  A hand-out is an
    array whose `base` chain ends in a _Lease token that only that array and its descendants reference; the pool itself
    keeps no reference to the token, only a weakref.finalize on it.  A block goes back to the idle list when - and only when -
    its token has been collected, i.e. when no array, view, recarray column, memoryview or np.frombuffer of the hand-out
    is alive.  What Python cannot see it cannot protect: a RAW ADDRESS taken from a result (arr.ctypes.data, a pointer kept
    by a C extension) does not keep the token alive - whoever holds one must keep the array too, or copy.
    Thread safety: empty() runs under a lock from the scan to the finished view; finalizers (which may run on any thread,
    also inside empty() through the garbage collector) only append to a deque, which empty() drains under the lock.
    Bounds: the pool never tracks more than `cap` bytes (leased + idle; $DV_HOST_POOL_GB, default min(24 GB, a quarter
    of the machine's RAM); 0 disables the pool) - beyond it a request gets a plain np.empty; idle blocks are kept only up
    to the total of the last four requests (a deblend_field call asks for three arrays) and are dropped when they have
    not been used for eight requests, so one large call does not pin its memory for the life of the process.
    host_pool_clear() drops every idle block at once.  Arrays below 64 MB are plain np.empty."""

    MIN_BYTES = 64 << 20
    KEEP_REQUESTS = 4          # idle bytes kept <= the total of this many most recent requests
    MAX_IDLE_AGE = 8           # ... and an idle block older than this many requests is dropped

    def __init__(self, cap_bytes: Optional[int] = None):
        if cap_bytes is None:
            try:
                cap_bytes = int(float(os.environ["DV_HOST_POOL_GB"]) * (1 << 30))
            except (KeyError, ValueError):
                cap_bytes = min(24 << 30, self._ram_bytes() // 4)
        self.cap = max(0, cap_bytes)
        self._lock = threading.Lock()
        self._idle: List[tuple] = []                       # (block, tick of its last use), least recently used first
        self._returned = collections.deque()               # blocks whose lease has died, not yet back on the idle list
        self._leased = 0                                   # bytes handed out and not yet returned
        self._recent = collections.deque(maxlen=self.KEEP_REQUESTS)
        self._tick = 0

    @staticmethod
    def _ram_bytes() -> int:
        try:
            return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
        except (ValueError, OSError, AttributeError):
            return 64 << 30

    def _give_back(self, block: np.ndarray):
        """finalizer of a lease (any thread, possibly inside empty() via the garbage collector): no lock taken here"""
        self._returned.append(block)

    def _drain(self):
        while True:
            try:
                block = self._returned.popleft()
            except IndexError:
                return
            self._leased -= block.nbytes
            self._idle.append((block, self._tick))

    def _trim(self, keep_bytes: int):
        """drops idle blocks that are too old, then the least recently used ones until at most keep_bytes stay idle"""
        self._idle = [(b, t) for b, t in self._idle if self._tick - t <= self.MAX_IDLE_AGE]
        total = sum(b.nbytes for b, _ in self._idle)
        while self._idle and total > keep_bytes:
            total -= self._idle.pop(0)[0].nbytes

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.MIN_BYTES or nbytes > self.cap:
            return np.empty(shape, dtype)
        with self._lock:
            self._tick += 1
            self._drain()
            self._recent.append(nbytes)
            pick = -1
            for i, (b, _) in enumerate(self._idle):           # best fit among the idle blocks
                if b.nbytes >= nbytes and (pick < 0 or b.nbytes < self._idle[pick][0].nbytes):
                    pick = i
            if pick >= 0:
                block = self._idle.pop(pick)[0]
            else:
                idle = sum(b.nbytes for b, _ in self._idle)
                if self._leased + idle + nbytes > self.cap:   # make room among the idle blocks first
                    self._trim(max(0, self.cap - self._leased - nbytes))
                    idle = sum(b.nbytes for b, _ in self._idle)
                if self._leased + idle + nbytes > self.cap:
                    return np.empty(shape, dtype)             # everything tracked is in use: a plain array, not tracked
                block = np.empty(nbytes, np.uint8)
            self._leased += block.nbytes
            self._trim(sum(self._recent))
            lease = _Lease(block, nbytes)
            weakref.finalize(lease, self._give_back, block)
            out = np.asarray(lease).view(dtype).reshape(shape)  # the only references to `lease` are out's base chain
            del lease
            return out

    def stats(self) -> dict:
        with self._lock:
            self._drain()
            return {"cap": self.cap, "leased_bytes": self._leased, "idle_bytes": sum(b.nbytes for b, _ in self._idle),
                    "idle_blocks": len(self._idle)}

    def clear(self):
        """drops every idle block (leased ones belong to their holders and are freed when those let go)"""
        with self._lock:
            self._drain()
            self._idle = []


_host_pool = _HostPool()


def host_pool_clear():
    """Frees the idle blocks of the result-array pool (see _HostPool) - e.g. after the last deblend_field of a session."""
    _host_pool.clear()


def host_pool_stats() -> dict:
    return _host_pool.stats()


def _f32c(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


class Context:
    """One per process: selects the GPU, owns the HIP streams and the RCCL communicator.

    Lifetime: every Engine registers with its Context; close() destroys the live engines first (their buffers, events
    and pinned memory hang off the context's device and streams), then the communicator and the streams.  Contexts still
    open when the interpreter exits are closed by an atexit hook, i.e. BEFORE Python finalises modules in arbitrary
    order and before the C runtime runs the HIP runtime's own exit handlers - relying on __del__ there left models
    alive behind their context (SIGSEGV inside exit() under rocprofv3)."""

    _live: "weakref.WeakSet[Context]" = weakref.WeakSet()

    def __init__(self, device: int = 0, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None):
        self._h = C.c_void_p()
        self._engines: "weakref.WeakSet[Engine]" = weakref.WeakSet()
        self.group = None                 # host-side rendezvous of the job (debvader_amd.parallel.make_context)
        self._owns_group = False
        self.rank, self.world, self.device = rank, world, device
        idbuf = None
        if world > 1:
            if unique_id is None or len(unique_id) != _lib.DV_UNIQUE_ID_BYTES:
                raise ValueError("world > 1 needs rank 0's unique id (Context.unique_id())")
            idbuf = C.create_string_buffer(unique_id, _lib.DV_UNIQUE_ID_BYTES)
        check(lib.dv_ctx_create(device, rank, world, idbuf, C.byref(self._h)))
        Context._live.add(self)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.DV_UNIQUE_ID_BYTES)
        check(lib.dv_comm_unique_id(buf))
        return buf.raw

    def sync(self):
        check(lib.dv_ctx_sync(self._h))

    def comm_info(self) -> Dict:
        """What the RCCL communicator of this context really spans: {comm_ranks (ncclCommCount, 0 without one), comm_rank,
        device, bus_id (PCI), rehearsal (DV_DEBUG_FAKE_PEERS: a one-rank communicator although world > 1)}."""
        n, r, d, reh = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        bus = C.create_string_buffer(32)
        check(lib.dv_ctx_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d), bus, 32, C.byref(reh)))
        return dict(comm_ranks=n.value, comm_rank=r.value, device=d.value, bus_id=bus.value.decode(),
                    rehearsal=bool(reh.value), world=self.world, rank=self.rank)

    def comm_prof(self, on: bool):
        """Time every collective on the comm stream and every wait of the main stream for one (dv_comm_prof_enable)."""
        check(lib.dv_comm_prof_enable(self._h, int(on)))

    def comm_prof_read(self) -> Dict:
        """{collectives, comm_ms, waits, exposed_ms} since the last read (synchronises)."""
        n1, n2, a, b = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        check(lib.dv_comm_prof_read(self._h, C.byref(n1), C.byref(a), C.byref(n2), C.byref(b)))
        return dict(collectives=n1.value, comm_ms=a.value, waits=n2.value, exposed_ms=b.value)

    def allreduce(self, values: Sequence[float]) -> np.ndarray:
        a = np.ascontiguousarray(values, dtype=np.float32)
        check(lib.dv_ctx_allreduce_host(self._h, _fp(a), a.size))
        return a

    # -- scene compositing (float64 host arrays, like the reference's numpy fields) --------------
    def scene_extract(self, field, starts, cutout_size: int) -> np.ndarray:
        """cutouts[i] = field[starts[i,0]:+cs, starts[i,1]:+cs, :] for a field (F, F, bands)."""
        field = np.ascontiguousarray(field, dtype=np.float64)
        starts = _i32_rows(starts, "cutout starts")
        if field.ndim != 3 or field.shape[0] != field.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {field.shape}")
        out = np.empty((starts.shape[0], cutout_size, cutout_size, field.shape[2]), np.float64)
        dp = C.POINTER(C.c_double)
        check(lib.dv_scene_extract(self._h, field.ctypes.data_as(dp), field.shape[0], field.shape[2],
                                   starts.ctypes.data_as(C.POINTER(C.c_int32)), starts.shape[0], int(cutout_size),
                                   out.ctypes.data_as(dp)))
        return out

    def scene_composite(self, field, stamps, positions, sign: float = 1.0) -> np.ndarray:
        """field + sign * sum_i shift(pad(stamps[i]), positions[i]) (scipy.ndimage.shift semantics), in object order."""
        out = np.array(field, dtype=np.float64, order="C", copy=True)
        stamps = np.ascontiguousarray(stamps, dtype=np.float64)
        positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        if out.ndim != 3 or out.shape[0] != out.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {out.shape}")
        if stamps.shape[0] == 0:
            return out
        if stamps.ndim != 4 or stamps.shape[1] != stamps.shape[2] or stamps.shape[3] != out.shape[2] \
                or stamps.shape[0] != positions.shape[0]:
            raise ValueError(f"stamps {stamps.shape} / positions {positions.shape} do not fit field {out.shape}")
        dp = C.POINTER(C.c_double)
        check(lib.dv_scene_composite(self._h, out.ctypes.data_as(dp), out.shape[0], out.shape[2],
                                     stamps.ctypes.data_as(dp), positions.ctypes.data_as(dp), stamps.shape[0],
                                     stamps.shape[1], float(sign)))
        return out

    def close(self):
        """Destroys the engines created on this context, then the context.  Idempotent."""
        for eng in list(self._engines):
            eng.close()
        if self._h:
            lib.dv_ctx_destroy(self._h)
            self._h = C.c_void_p()
        if self.group is not None and self._owns_group:
            try:
                self.group.close()
            except Exception:
                pass
        self.group = None
        Context._live.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@atexit.register
def _close_all_contexts():
    global _default_ctx
    for ctx in list(Context._live):
        try:
            ctx.close()
        except Exception:
            pass
    _default_ctx = None


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0, 0, 1, None)
    return _default_ctx


def set_default_context(ctx: Optional[Context]):
    global _default_ctx
    _default_ctx = ctx


class Engine:
    """Weights, Adam state, workspaces and step functions of one conv-VAE on one GPU."""

    def __init__(self, cfg: DvConfig, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self.cfg = cfg
        self._h = C.c_void_p()
        if not self.ctx._h:
            raise RuntimeError("the Context of this Engine has been closed")
        check(lib.dv_model_create(self.ctx._h, C.byref(cfg), C.byref(self._h)))
        self.ctx._engines.add(self)
        self.specs = arch_specs(cfg)
        self.index = {n: i for i, (n, _, _) in enumerate(self.specs)}
        self.latent = cfg.latent_dim
        self.tw = self.latent + self.latent * (self.latent + 1) // 2
        self.stamp_shape = (cfg.height, cfg.width, cfg.bands)
        self.max_batch = cfg.max_batch

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self._h:
            lib.dv_model_destroy(self._h)
            self._h = C.c_void_p()
        try:
            self.ctx._engines.discard(self)
        except Exception:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters -------------------------------------------------------------------------
    def init(self, seed: int = 0):
        check(lib.dv_model_init(self._h, seed))

    def _tensor(self, fn, i, *extra) -> np.ndarray:
        name, shape, _ = self.specs[i]
        a = np.empty(shape, dtype=np.float32)
        check(fn(self._h, i, *extra, _fp(a), a.nbytes))
        return a

    def get_param(self, key) -> np.ndarray:
        return self._tensor(lib.dv_model_get_param, self._idx(key))

    def get_grad(self, key) -> np.ndarray:
        return self._tensor(lib.dv_model_get_grad, self._idx(key))

    def get_slot(self, key, which: int) -> np.ndarray:
        return self._tensor(lib.dv_model_get_slot, self._idx(key), which)

    def set_param(self, key, value):
        i = self._idx(key)
        a = _f32c(value, self.specs[i][1])
        check(lib.dv_model_set_param(self._h, i, _fp(a), a.nbytes))

    def set_slot(self, key, which: int, value):
        i = self._idx(key)
        a = _f32c(value, self.specs[i][1])
        check(lib.dv_model_set_slot(self._h, i, which, _fp(a), a.nbytes))

    def _idx(self, key) -> int:
        return key if isinstance(key, int) else self.index[key]

    def get_params(self) -> Dict[str, np.ndarray]:
        return {n: self.get_param(i) for i, (n, _, _) in enumerate(self.specs)}

    def set_params(self, params: Dict[str, np.ndarray]):
        for n, v in params.items():
            self.set_param(n, v)

    def set_trainable(self, encoder: bool, decoder: bool):
        check(lib.dv_model_set_trainable(self._h, int(encoder), int(decoder)))

    def optimizer_reset(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7):
        check(lib.dv_optimizer_reset(self._h, lr, beta1, beta2, eps))

    @property
    def iterations(self) -> int:
        it = C.c_int64()
        check(lib.dv_optimizer_get_iter(self._h, C.byref(it)))
        return it.value

    @iterations.setter
    def iterations(self, v: int):
        check(lib.dv_optimizer_set_iter(self._h, int(v)))

    # -- data -------------------------------------------------------------------------------
    def upload(self, slot: int, x, y):
        x = _f32c(x)
        y = _f32c(y)
        if x.shape != y.shape or x.shape[1:] != self.stamp_shape:
            raise ValueError(f"expected x, y of shape (N,{self.stamp_shape}), got {x.shape} and {y.shape}")
        check(lib.dv_data_upload(self._h, slot, _fp(x), _fp(y), x.shape[0]))
        return x.shape[0]

    def free_data(self, slot: int):
        check(lib.dv_data_free(self._h, slot))

    # -- steps ------------------------------------------------------------------------------
    def _step(self, fn, slot, idx, first, B, global_batch, eps, seed) -> Dict[str, float]:
        out = np.zeros(_lib.DV_N_SCALARS, dtype=np.float32)
        ip = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            B = idx.size
            ip = idx.ctypes.data_as(C.POINTER(C.c_int32))
        if eps is not None:
            eps = _f32c(eps, (B, self.latent))
        check(fn(self._h, slot, ip, int(first), int(B), int(global_batch or 0), _fp(eps), int(seed), _fp(out)))
        return {k: float(out[i]) for i, k in enumerate(_lib.SCALAR_NAMES)}

    def train_step(self, slot=0, idx=None, first=0, B=None, global_batch=None, eps=None, seed=0):
        return self._step(lib.dv_train_step, slot, idx, first, B, global_batch, eps, seed)

    def eval_step(self, slot=1, idx=None, first=0, B=None, global_batch=None, eps=None, seed=0):
        return self._step(lib.dv_eval_step, slot, idx, first, B, global_batch, eps, seed)

    def grad_step(self, slot=0, idx=None, first=0, B=None, global_batch=None, eps=None, seed=0):
        return self._step(lib.dv_grad_step, slot, idx, first, B, global_batch, eps, seed)

    def train_step_async(self, ticket, slot=0, idx=None, first=0, B=None, global_batch=None, seed=0):
        """Queues a training step under `ticket` (0..3); step_result(ticket) returns its scalars later."""
        ip = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            B = idx.size
            ip = idx.ctypes.data_as(C.POINTER(C.c_int32))
        check(lib.dv_train_step_async(self._h, slot, ip, int(first), int(B), int(global_batch or 0), int(seed), int(ticket)))

    def step_result(self, ticket) -> Dict[str, float]:
        out = np.zeros(_lib.DV_N_SCALARS, dtype=np.float32)
        check(lib.dv_step_result(self._h, int(ticket), _fp(out)))
        return {k: float(out[i]) for i, k in enumerate(_lib.SCALAR_NAMES)}

    def train_steps(self, slot, first, B, steps, global_batch=None, seed=0) -> Dict[str, float]:
        out = np.zeros(_lib.DV_N_SCALARS, dtype=np.float32)
        check(lib.dv_train_steps(self._h, slot, int(first), int(B), int(global_batch or 0), int(steps), int(seed),
                                 _fp(out)))
        return {k: float(out[i]) for i, k in enumerate(_lib.SCALAR_NAMES)}

    # -- inference --------------------------------------------------------------------------
    def set_normalise(self, on: bool):
        """tanh(arcsinh) on inference inputs and the inverse on the predicted mean, on the GPU (deblend(normalise=True))."""
        check(lib.dv_model_set_normalise(self._h, 1 if on else 0))

    def set_mse_sample(self, on: bool):
        """The "mse" scalar of the step functions against a SAMPLE of the output distribution (Keras semantics of the
        reference's compile(metrics=["mse"]), model.py:158) instead of its mean."""
        check(lib.dv_model_set_mse_sample(self._h, 1 if on else 0))

    def keep_outputs(self, on: bool):
        """Gradient / train steps also write loc and scale of their forward pass (activation("loc"), activation("scale"));
        off by default - a train step has no reader for them."""
        check(lib.dv_model_set_keep_outputs(self._h, 1 if on else 0))

    def infer(self, x, eps=None, seed=0, want=("loc", "scale"), out=None) -> Dict[str, np.ndarray]:
        """One stochastic forward pass over all stamps.  float64 arrays (numpy's default, what the reference's callers
        pass) go to the engine as they are: the float32 cast of deblender.py:18 happens while the library stages them."""
        x = np.asarray(x)
        f64 = x.dtype == np.float64 and x.flags.c_contiguous
        if not f64:
            x = _f32c(x)
        if x.ndim != 4 or x.shape[1:] != self.stamp_shape:
            raise ValueError(f"expected images of shape (N,{self.stamp_shape}), got {x.shape}")
        N = x.shape[0]
        bufs = {}
        for k in ("loc", "scale", "mu", "zstd", "z"):
            shape = (N,) + self.stamp_shape if k in ("loc", "scale") else (N, self.latent)
            if k not in want:
                bufs[k] = None
            elif out is not None and k in out:      # caller-provided result array (reused across calls)
                if out[k].shape != shape or out[k].dtype != np.float32 or not out[k].flags.c_contiguous:
                    raise ValueError(f"out[{k!r}] must be a C-contiguous float32 array of shape {shape}")
                bufs[k] = out[k]
            else:
                bufs[k] = _host_pool.empty(shape, np.float32)
        if eps is not None:
            eps = _f32c(eps, (N, self.latent))
        if f64:
            check(lib.dv_infer_f64(self._h, x.ctypes.data_as(C.POINTER(C.c_double)), N, _fp(eps), int(seed),
                                   _fp(bufs["loc"]), _fp(bufs["scale"]), _fp(bufs["mu"]), _fp(bufs["zstd"]), _fp(bufs["z"])))
        else:
            check(lib.dv_infer(self._h, _fp(x), N, _fp(eps), int(seed), _fp(bufs["loc"]), _fp(bufs["scale"]),
                               _fp(bufs["mu"]), _fp(bufs["zstd"]), _fp(bufs["z"])))
        return {k: v for k, v in bufs.items() if v is not None}

    def infer_cutouts(self, field, starts, seed=0, want=("loc", "scale"), out=None) -> Dict[str, np.ndarray]:
        """infer() on the cutouts field[x:x+H, y:y+H, :] of a float64 field (F, F, bands) for every row (x, y) of `starts`,
        gathered and cast on the GPU (dv_infer_cutouts): the stamps never visit the host.  Bit-identical to
        infer(ctx.scene_extract(field, starts, H)) with the same seed."""
        field = np.ascontiguousarray(field, dtype=np.float64)
        starts = _i32_rows(starts, "cutout starts")
        if field.ndim != 3 or field.shape[0] != field.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {field.shape}")
        N = starts.shape[0]
        bufs = {}
        for k in ("loc", "scale", "mu", "zstd", "z"):
            shape = (N,) + self.stamp_shape if k in ("loc", "scale") else (N, self.latent)
            if k not in want:
                bufs[k] = None
            elif out is not None and k in out:
                if out[k].shape != shape or out[k].dtype != np.float32 or not out[k].flags.c_contiguous:
                    raise ValueError(f"out[{k!r}] must be a C-contiguous float32 array of shape {shape}")
                bufs[k] = out[k]
            else:
                bufs[k] = _host_pool.empty(shape, np.float32)
        check(lib.dv_infer_cutouts(self._h, field.ctypes.data_as(C.POINTER(C.c_double)), field.shape[0], field.shape[2],
                                   starts.ctypes.data_as(C.POINTER(C.c_int32)), N, int(seed), _fp(bufs["loc"]),
                                   _fp(bufs["scale"]), _fp(bufs["mu"]), _fp(bufs["zstd"]), _fp(bufs["z"])))
        return {k: v for k, v in bufs.items() if v is not None}

    def infer_cutouts_keep(self, field, starts, seed=0) -> Dict[str, np.ndarray]:
        """infer_cutouts() for a caller that also needs the float64 cutouts (dv_infer_cutouts_keep): returns
        {"loc", "scale" (float32), "cutouts" (float64)}, all (N,) + stamp shape.  The cutouts are assembled on the host from the
        field the caller holds while the GPU runs the forward passes - bit-identical to ctx.scene_extract(field, starts, H) -
        and mean / stddev are bit-identical to infer(cutouts) with the same seed."""
        field = np.ascontiguousarray(field, dtype=np.float64)
        starts = _i32_rows(starts, "cutout starts")
        if field.ndim != 3 or field.shape[0] != field.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {field.shape}")
        N = starts.shape[0]
        out = {"loc": _host_pool.empty((N,) + self.stamp_shape, np.float32),
               "scale": _host_pool.empty((N,) + self.stamp_shape, np.float32),
               "cutouts": _host_pool.empty((N,) + self.stamp_shape, np.float64)}
        check(lib.dv_infer_cutouts_keep(self._h, field.ctypes.data_as(C.POINTER(C.c_double)), field.shape[0], field.shape[2],
                                        starts.ctypes.data_as(C.POINTER(C.c_int32)), N, int(seed), _fp(out["loc"]),
                                        _fp(out["scale"]), out["cutouts"].ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def infer_cutouts_composite(self, field, starts, places, seed=0, residual=True, mse_center=True) -> Dict[str, np.ndarray]:
        """infer_cutouts() with the compositing that follows it in the reference done on the GPU (dv_infer_cutouts_composite):
        returns {"mean_field", "stddev_field", ["residual_field"], ["mse_center"]} - float64 (F, F, bands) sums of the
        network's mean / stddev stamps placed at `places` (row, col of each stamp's top-left corner; off-field parts are
        dropped) in object order, the field minus the mean stamps, and each stamp's centre-10x10 MSE against its cutout.
        No stamp visits the host."""
        field = np.ascontiguousarray(field, dtype=np.float64)
        starts = _i32_rows(starts, "cutout starts")
        places = _i32_rows(places, "stamp placements")
        if field.ndim != 3 or field.shape[0] != field.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {field.shape}")
        if places.shape != starts.shape:
            raise ValueError(f"{starts.shape[0]} cutout starts but {places.shape[0]} placements")
        N = starts.shape[0]
        dp = C.POINTER(C.c_double)
        out = {"mean_field": np.empty(field.shape, np.float64), "stddev_field": np.empty(field.shape, np.float64)}
        if residual:
            out["residual_field"] = np.empty(field.shape, np.float64)
        if mse_center:
            out["mse_center"] = np.empty((N,), np.float64)
        opt = lambda k: out[k].ctypes.data_as(dp) if k in out else None
        check(lib.dv_infer_cutouts_composite(self._h, field.ctypes.data_as(dp), field.shape[0], field.shape[2],
                                             starts.ctypes.data_as(C.POINTER(C.c_int32)),
                                             places.ctypes.data_as(C.POINTER(C.c_int32)), N, int(seed),
                                             out["mean_field"].ctypes.data_as(dp), out["stddev_field"].ctypes.data_as(dp),
                                             opt("residual_field"), opt("mse_center")))
        return out

    def infer_cutouts_stream(self, field, starts, consumer, seed=0):
        """infer_cutouts() for inputs whose outputs do not belong on one host (a million cutouts: 167 GB): every finished
        chunk is handed to consumer(first, mean, stddev) - float32 views (count, H, H, bands) of the pinned transfer
        buffers, valid until the consumer returns, stamps [first, first + count) in input order.  Nothing is copied on
        the host; the GPU works on the next chunks meanwhile."""
        field = np.ascontiguousarray(field, dtype=np.float64)
        starts = _i32_rows(starts, "cutout starts")
        if field.ndim != 3 or field.shape[0] != field.shape[1]:
            raise ValueError(f"expected a square field (F, F, bands), got {field.shape}")
        failure = []

        def trampoline(_user, first, count, mean_p, std_p):
            try:
                shape = (int(count),) + self.stamp_shape
                consumer(int(first), np.ctypeslib.as_array(mean_p, shape=shape), np.ctypeslib.as_array(std_p, shape=shape))
                return 0
            except BaseException as e:          # never let an exception cross the C frames
                failure.append(e)
                return 1

        cb = _lib.CHUNK_FN(trampoline)
        rc = lib.dv_infer_cutouts_stream(self._h, field.ctypes.data_as(C.POINTER(C.c_double)), field.shape[0],
                                         field.shape[2], starts.ctypes.data_as(C.POINTER(C.c_int32)), starts.shape[0],
                                         int(seed), C.cast(cb, C.c_void_p), None)
        if failure:
            raise failure[0]
        check(rc)

    def infer_mc(self, x, nsamples=100, seed=0):
        """(mean, std) over `nsamples` stochastic decodes of every stamp (encoder runs once per stamp)."""
        x = _f32c(x)
        if x.ndim != 4 or x.shape[1:] != self.stamp_shape:
            raise ValueError(f"expected images of shape (N,{self.stamp_shape}), got {x.shape}")
        mean = np.empty(x.shape, np.float32)
        std = np.empty(x.shape, np.float32)
        check(lib.dv_infer_mc(self._h, _fp(x), x.shape[0], int(nsamples), int(seed), _fp(mean), _fp(std)))
        return mean, std

    def encode(self, x) -> np.ndarray:
        x = _f32c(x)
        if x.ndim != 4 or x.shape[1:] != self.stamp_shape:
            raise ValueError(f"expected images of shape (N,{self.stamp_shape}), got {x.shape}")
        t = np.empty((x.shape[0], self.tw), np.float32)
        check(lib.dv_encode(self._h, _fp(x), x.shape[0], _fp(t)))
        return t

    def decode(self, z) -> Tuple[np.ndarray, np.ndarray]:
        z = _f32c(z)
        if z.ndim != 2 or z.shape[1] != self.latent:
            raise ValueError(f"expected latents of shape (N,{self.latent}), got {z.shape}")
        loc = np.empty((z.shape[0],) + self.stamp_shape, np.float32)
        scale = np.empty_like(loc)
        check(lib.dv_decode(self._h, _fp(z), z.shape[0], _fp(loc), _fp(scale)))
        return loc, scale

    # -- introspection ----------------------------------------------------------------------
    def activation(self, name: str, shape) -> np.ndarray:
        a = np.empty(shape, np.float32)
        check(lib.dv_model_get_activation(self._h, name.encode(), _fp(a), a.nbytes))
        return a

    def prof_enable(self, on: bool):
        check(lib.dv_prof_enable(self._h, int(on)))

    def prof_reset(self):
        check(lib.dv_prof_reset(self._h))

    def prof_families(self) -> List[Dict]:
        """[{name, launches, ms, flops, executed_flops, algorithmic_bytes}] per MFMA kernel family since the last prof_reset
        (names as rocprofv3 prints them; executed_flops: what the matrix pipe executes - Winograd kernels execute fewer
        than the direct-convolution count they are priced with)."""
        out = []
        fam = 0
        while True:
            name = C.create_string_buffer(96)
            n, ms, fl, ex, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
            if lib.dv_prof_read_family(self._h, fam, name, 96, C.byref(n), C.byref(ms), C.byref(fl), C.byref(ex),
                                       C.byref(by)) != 0:
                break
            if n.value:
                out.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, flops=fl.value,
                                executed_flops=ex.value, algorithmic_bytes=by.value))
            fam += 1
        return out

    def prof_read(self, klass: int) -> Tuple[int, float]:
        n, ms = C.c_int64(), C.c_double()
        check(lib.dv_prof_read(self._h, klass, C.byref(n), C.byref(ms)))
        return n.value, ms.value
