"""debvader_amd.parallel.HostGroup: the torch-free rendezvous of a multi-rank job (RCCL id broadcast, barriers, max over
ranks, object gather), three processes on CPU; and parallel.make_context driven through it with the stub engine."""
import multiprocessing as mp
import os
import socket
import struct
import tempfile

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, env, q):
    try:
        os.environ.update(env)
        if not os.environ.get("DV_RDZV_TOKEN"):
            os.environ.pop("DV_RDZV_TOKEN", None)
        from debvader_amd import parallel

        with parallel.HostGroup(rank, world, timeout=60) as g:
            uid = g.broadcast(bytes(range(128)) if rank == 0 else None)
            parts = g.allgather(struct.pack("<I", rank) * (rank + 1))            # ragged payloads, rank order
            g.barrier()
            mx = g.max(1.0 + rank)
            objs = g.gather_object({"rank": rank, "data": list(range(rank))}, dst=0)
            late = g.broadcast(b"from-2" if rank == 2 else None, src=2)
        q.put((rank, uid, parts, mx, objs, late, None))
    except Exception as e:                                                        # pragma: no cover
        q.put((rank, None, None, None, None, None, repr(e)))


@pytest.mark.parametrize("mode", ["master_port", "port_file", "env_token"])
def test_three_ranks_meet_and_exchange(mode, monkeypatch):
    world = 3
    port = _free_port()
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "DV_RDZV_TOKEN": ""}
    if mode == "env_token":                     # the launcher exports the secret: no file at all
        env["DV_RDZV_TOKEN"] = "ab" * 16
    holder = None
    stale = None
    if mode == "port_file":
        # what torch.distributed.run does: its agent keeps MASTER_PORT busy and tells the workers so
        holder = socket.socket()
        holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        holder.bind(("127.0.0.1", port))
        holder.listen(1)
        env["TORCHELASTIC_USE_AGENT_STORE"] = "True"
        # a stale file of an earlier job with the same launcher pid / port: a dead port and another token
        from debvader_amd import parallel

        stale = os.path.join(parallel._private_dir(), f"job_{port}_{os.getpid()}")
        with open(stale, "w") as fh:
            fh.write(f"{_free_port()} {'00' * 16}\n")
    else:
        env["TORCHELASTIC_USE_AGENT_STORE"] = ""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, env, q)) for r in range(world)]
    # rank 0 last: the spokes must wait (and, with a stale port file, retry) until the hub is up
    for p in procs[1:] + procs[:1]:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    if holder is not None:
        holder.close()
    for rank, uid, parts, mx, objs, late, err in res:
        assert err is None, err
        assert uid == bytes(range(128))
        assert parts == [struct.pack("<I", r) * (r + 1) for r in range(world)]
        assert mx == 3.0 and late == b"from-2"
        if rank == 0:
            assert objs == [{"rank": r, "data": list(range(r))} for r in range(world)]
        else:
            assert objs is None
    if stale is not None:
        assert not os.path.exists(stale)                    # rank 0 replaced it and removed its own on close


def _ctx_worker(rank, world, env, log, q):
    try:
        os.environ.update(env, DV_STUB_LOG=log)
        import importlib

        from debvader_amd import parallel

        stub = importlib.import_module("tests.stub_engine")
        parallel.E = stub                                   # Context / unique_id of the stand-in (no GPU here)
        ctx = parallel.make_context(rank, world, local_rank=rank)
        assert ctx.rank == rank and ctx.world == world and ctx.group is not None
        ctx.group.barrier()
        ctx.group.close()
        q.put((rank, None))
    except Exception as e:                                  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 8])
def test_make_context_hands_rank0s_id_to_every_rank(tmp_path, world):
    import json

    log = str(tmp_path / "stub")
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TORCHELASTIC_USE_AGENT_STORE": ""}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ctx_worker, args=(r, world, env, log, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert [e for _, e in res] == [None] * world, res
    uids, devs = [], []
    for r in range(world):
        ev = [json.loads(ln) for ln in open(f"{log}.{r}")]
        uids.append(next(e["uid"] for e in ev if e["event"] == "ctx"))
        devs.append(next(e["device"] for e in ev if e["event"] == "ctx"))
    assert len(set(uids)) == 1 and len(uids[0]) == 256 and devs == list(range(world))


def _clash_worker(rank, world, env, log, q):
    try:
        os.environ.update(env, DV_STUB_LOG=log, DV_STUB_BUS="0000:05:00.0")       # every rank names the same GPU
        import importlib

        from debvader_amd import parallel

        parallel.E = importlib.import_module("tests.stub_engine")
        try:
            parallel.make_context(rank, world, local_rank=0)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    except Exception as e:                                  # pragma: no cover
        q.put((rank, repr(e)))


def test_two_ranks_mapped_onto_one_gpu_fail_fast_on_every_rank(tmp_path):
    """LOCAL_RANK beyond the visible devices used to wrap around (two ranks on one GPU, an error or a stall inside
    ncclCommInitRank later).  Now the ranks compare (host, PCI bus id) through the host group BEFORE the communicator is built
    and every rank raises with the same message; no engine context is created."""
    import json

    world, log = 2, str(tmp_path / "stub")
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TORCHELASTIC_USE_AGENT_STORE": ""}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_clash_worker, args=(r, world, env, log, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    for _, msg in res:
        assert "ranks 0 and 1" in msg and "0000:05:00.0" in msg and "one GPU per rank" in msg, res
    assert not any(os.path.exists(f"{log}.{r}") for r in range(world))         # no Context was ever constructed


def test_local_rank_beyond_the_visible_gpus_is_an_error_not_a_modulo(monkeypatch):
    from debvader_amd import parallel

    class _E:
        @staticmethod
        def device_count():
            return 4

    monkeypatch.setattr(parallel, "E", _E)
    with pytest.raises(RuntimeError, match="only 4 GPUs are visible"):
        parallel.make_context(0, 1, local_rank=5)


def _beyond_worker(rank, world, env, log, q):
    try:
        os.environ.update(env, DV_STUB_LOG=log, DV_STUB_VISIBLE="4")
        import importlib
        import threading

        from debvader_amd import parallel

        parallel.E = importlib.import_module("tests.stub_engine")
        try:
            parallel.make_context(rank, world, local_rank=(5 if rank == 1 else 0))     # rank 1 cannot open its GPU
            q.put((rank, "no error", 0))
        except RuntimeError as e:
            q.put((rank, str(e), sum(t.name.startswith("dv-hub") for t in threading.enumerate())))
    except Exception as e:                                  # pragma: no cover
        q.put((rank, repr(e), -1))


def test_a_rank_that_cannot_open_its_gpu_fails_every_rank_together(tmp_path):
    """the LOCAL_RANK >= visible-GPUs check used to raise on that rank alone, BEFORE it joined the host group -
    its healthy peers then sat in the rendezvous until the timeout.  Now every rank joins first, the unlucky one says what
    is wrong in the device all-gather, ALL ranks raise the same message within seconds, nobody builds an engine context
    and a group that make_context created itself is closed again."""
    import time

    world, log = 2, str(tmp_path / "stub")
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TORCHELASTIC_USE_AGENT_STORE": "",
           "DV_RDZV_TIMEOUT": "60"}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=_beyond_worker, args=(r, world, env, log, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert time.time() - t0 < 50                                               # (nobody waited for the rendezvous timeout)
    for _, msg, _ in res:
        assert "rank 1: LOCAL_RANK 5 but only 4 GPUs are visible" in msg, res
    assert not any(os.path.exists(f"{log}.{r}") for r in range(world))         # no Context was ever constructed


def _hub_only(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_USE_AGENT_STORE="")
    os.environ.pop("DV_RDZV_TOKEN", None)
    from debvader_amd import parallel

    try:
        parallel.HostGroup(0, 2, timeout=3)
        q.put("joined")
    except TimeoutError:
        q.put("timeout")


def test_a_stranger_on_master_port_cannot_join_as_a_rank():
    """in the direct mode (mpirun / srun / a shell loop: rank 0 listens on MASTER_PORT itself) any process that
    could reach the port used to be admitted, and rank 0 then unpickled its bytes.  Now every handshake carries the job's
    secret; a hello with the right magic, world and rank but without it is refused, and the hub times out cleanly
    (listener closed, token file gone)."""
    from debvader_amd import parallel

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    hub = ctx.Process(target=_hub_only, args=(port, q))
    hub.start()
    answer = None
    import time
    for _ in range(100):
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            s.sendall(parallel._MAGIC + struct.pack("<II", 1, 2) + b"\0" * 16)
            answer = s.recv(2)
            s.close()
            break
        except OSError:
            time.sleep(0.05)
    assert answer == b"NO"
    assert q.get(timeout=30) == "timeout"
    hub.join(30)
    assert not os.path.exists(os.path.join(parallel._private_dir(), f"job_{port}"))


def test_token_file_is_private_and_payloads_are_not_pickled(tmp_path, monkeypatch):
    from debvader_amd import parallel
    import numpy as np
    import stat

    d = parallel._private_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and (st.st_mode & 0o077) == 0 and st.st_uid == os.getuid()
    # a world-accessible directory in its place is refused, not used
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    bad = tmp_path / f"dv_rdzv_{os.getuid()}"
    bad.mkdir(mode=0o755)
    os.chmod(bad, 0o755)
    with pytest.raises(PermissionError):
        parallel._private_dir()
    # the wire format: plain types and numeric arrays only
    piece = (3, 9, np.arange(24, dtype=np.float32).reshape(2, 3, 4), np.zeros((0, 5), np.float64))
    back = parallel._unpack(parallel._pack(piece))
    assert back[:2] == (3, 9) and np.array_equal(back[2], piece[2]) and back[3].shape == (0, 5)
    with pytest.raises(TypeError):
        parallel._pack(object())
    with pytest.raises(TypeError):
        parallel._pack(np.array([object()]))
    import inspect
    assert "pickle" not in inspect.getsource(parallel).replace("never pickle", "").replace("never pickled", "").replace("pickle.loads on a socket", "").replace("unpickled", "")


def _gather_worker(rank, world, env, q):
    try:
        os.environ.update(env)
        os.environ.pop("DV_RDZV_TOKEN", None)
        import numpy as np
        from debvader_amd import parallel

        sent = []
        with parallel.HostGroup(rank, world, timeout=60) as g:
            if g._sock is not None:                       # count what a spoke RECEIVES during the gather
                real = g._sock.recv

                def counting(n, *a):
                    b = real(n, *a)
                    sent.append(len(b))
                    return b

                class Wrap:
                    def __init__(self, s): self._s = s
                    def recv(self, n, *a): return counting(n, *a)
                    def __getattr__(self, k): return getattr(self._s, k)
                g._sock = Wrap(g._sock)
            piece = (rank, rank + 1, np.full((1000, 7), float(rank), np.float32))
            got = g.gather_object(piece, dst=0)
            to2 = g.gather_object({"r": rank}, dst=2)
            g.barrier()
        q.put((rank, None if got is None else [(a, b, float(c.sum())) for a, b, c in got], to2, sum(sent), None))
    except Exception as e:                                # pragma: no cover
        q.put((rank, None, None, 0, repr(e)))


def test_gather_sends_pieces_to_the_hub_only():
    """gather_object was built on allgather - rank 0 sent the whole blob back to every spoke.  Now a spoke
    receives a 1-byte ack (and the small barrier frames), not world x 28 KB."""
    world = 3
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TORCHELASTIC_USE_AGENT_STORE": ""}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, env, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    for rank, got, to2, nrecv, err in res:
        assert err is None, err
        if rank == 0:
            assert got == [(r, r + 1, 7000.0 * r) for r in range(world)]
        else:
            assert got is None
            assert nrecv < 2000, nrecv                        # acks + barrier frames + (rank 2) three tiny dicts
        assert to2 == ([{"r": r} for r in range(world)] if rank == 2 else None)
