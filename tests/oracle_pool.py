"""Worker processes that evaluate the CPU oracle while the GPU tests run (TEST INFRASTRUCTURE).

The whole-step parity tests at the quoted batch sizes spend 40 - 70 s each in the numpy oracle (float64, float32 and
bf16-rounding evaluations of 256 stamps) - 240 of the GPU suite's 570 s with the GPU idle.  A session fixture
(tests/conftest.py) starts this pool BEFORE the first GPU call of the session and submits tests/oracle_jobs.py::HEAVY;
the tests that need those evaluations are moved to the end of the run, so the workers compute beside the rest of the suite.

Rules the pool keeps:
  * workers are SPAWNED (fresh interpreters: no HIP state is inherited, nothing is forked from a process that may touch
    the GPU later) and never import the engine;
  * one BLAS thread per worker (OPENBLAS / OMP / MKL_NUM_THREADS=1 in the child's environment): this image's OpenBLAS is not
    safe under concurrent callers inside ONE process - separate single-threaded processes are, and the
    results are bit-identical to the single-threaded inline evaluation;
  * a case that was not submitted (or a session without the pool: CPU runs, -k selections) is evaluated inline by the
    same function, so a test never depends on the pool for its result.
"""
import multiprocessing as mp
import os
import time

_pool = None
_pending = {}
_stats = {"waited_s": 0.0, "fetched": 0, "inline": 0}
_last_inline = {}


def _key(fn, kwargs):
    return fn + "|" + repr(sorted((k, repr(v)) for k, v in kwargs.items()))


def _call(fn, kwargs):
    from tests import oracle_jobs

    t0 = time.perf_counter()
    out = getattr(oracle_jobs, fn)(**kwargs)
    return out, time.perf_counter() - t0


def start(n_workers=None):
    """Spawns the workers and queues every HEAVY case.  Idempotent."""
    global _pool
    if _pool is not None:
        return
    from tests import oracle_jobs

    if n_workers is None:
        n_workers = max(1, min(len(oracle_jobs.HEAVY), (os.cpu_count() or 2) // 2, 6))
    saved = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update({k: "1" for k in saved})                     # inherited by the spawned children only
    try:
        _pool = mp.get_context("spawn").Pool(n_workers)
        for fn, kwargs in oracle_jobs.HEAVY:
            _pending[_key(fn, kwargs)] = _pool.apply_async(_call, (fn, kwargs))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def fetch(fn, **kwargs):
    """The result of tests.oracle_jobs.<fn>(**kwargs): from a worker when the case was queued, else computed here."""
    k = _key(fn, kwargs)
    res = _pending.pop(k, None)
    if res is None:
        if _last_inline.get("key") == k:                # the same case twice in a row (a test that repeats a case under
            return _last_inline["out"]                  # another kernel selection): evaluated once
        _stats["inline"] += 1
        out = _call(fn, kwargs)[0]
        _last_inline.update(key=k, out=out)
        return out
    t0 = time.perf_counter()
    out, _ = res.get(timeout=1800)
    _stats["waited_s"] += time.perf_counter() - t0
    _stats["fetched"] += 1
    return out


def stop():
    global _pool
    if _pool is not None:
        _pool.terminate()
        _pool.join()
        _pool = None
        _pending.clear()


def stats():
    return dict(_stats, queued=len(_pending), running=_pool is not None)
