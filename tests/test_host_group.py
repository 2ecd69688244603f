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


@pytest.mark.parametrize("mode", ["master_port", "port_file"])
def test_three_ranks_meet_and_exchange(mode, monkeypatch):
    world = 3
    port = _free_port()
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
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
        stale = os.path.join(tempfile.gettempdir(), f"dv_rdzv_{os.getuid()}_{os.getpid()}_{port}")
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


def test_make_context_hands_rank0s_id_to_every_rank(tmp_path):
    import json

    world, log = 2, str(tmp_path / "stub")
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TORCHELASTIC_USE_AGENT_STORE": ""}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ctx_worker, args=(r, world, env, log, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert [e for _, e in res] == [None, None], res
    uids = []
    for r in range(world):
        ev = [json.loads(ln) for ln in open(f"{log}.{r}")]
        uids.append(next(e["uid"] for e in ev if e["event"] == "ctx"))
    assert uids[0] == uids[1] and len(uids[0]) == 256
