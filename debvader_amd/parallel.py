"""Data-parallel plumbing: one process per GPU, ranks joined through RCCL inside the engine.

The host side only has to (1) hand rank 0's RCCL unique id to every rank, (2) split a global batch / an inference set
into contiguous per-rank shards and (3) offer the launcher-level barrier / max-over-ranks bench.py
needs.  All of it runs over a plain TCP star (`HostGroup`, rank 0 is the hub) found through the MASTER_ADDR /
MASTER_PORT / RANK / WORLD_SIZE variables every launcher exports: no torch (or any other framework) is imported into a
process that drives a GPU.  The reference is single-process (training/train.py:27-37); this layer is north_star's.
"""
from __future__ import annotations

import os
import pickle
import secrets
import socket
import struct
import tempfile
import time
from typing import List, Optional, Tuple

from . import engine as E


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of n items owned by `rank`: sizes differ by at most one, earlier ranks larger."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def loss_normalisers(global_batch: int, stamp_elems: int, kl_weight: float, kl_multiplicity: int):
    """Factors a shard applies so that per-shard sums ADD UP to the single-device Keras losses:
    nll_mean = sum(nll) / (B_global * H*W*C);  kl_reg = k * w * sum_b(KL_b) / B_global^2."""
    return 1.0 / (global_batch * stamp_elems), kl_multiplicity * kl_weight / float(global_batch) ** 2


# ---------------------------------------------------------------------------------------------------------------------
# host-side rendezvous
# ---------------------------------------------------------------------------------------------------------------------
_MAGIC = b"DVRDZV01"


def _send_msg(sock: socket.socket, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class HostGroup:
    """The ranks of one job as a TCP star with rank 0 in the middle: all-gather of small byte strings, and broadcast /
    barrier / max on top of it.  Meant for a handful of messages per run (RCCL id, barriers around a timed region, one
    float); everything that matters for speed goes through RCCL inside the engine.

    Where the ranks meet.  Launchers export MASTER_ADDR / MASTER_PORT.  `python -m torch.distributed.run` keeps its own
    store listening on that port for the life of the job (TORCHELASTIC_USE_AGENT_STORE=True), so there rank 0 listens on
    an ephemeral port of MASTER_ADDR and publishes it in a file the ranks of THIS launch can derive - temp dir, user id,
    launcher pid (their common parent) and MASTER_PORT - together with a random token that the handshake checks (a stale
    file of a dead job is refused or unanswered, and the client reads the file again).  Any other launcher (mpirun,
    srun, a shell loop) leaves MASTER_PORT free and rank 0 listens on it directly.  DV_RDZV_PORT forces a port.
    Single node, like the engine (one RCCL communicator over the GPUs of a box)."""

    def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None,
                 timeout: float = 300.0):
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        self.rank, self.world, self.timeout = rank, world, timeout
        self._conns: List[Optional[socket.socket]] = []
        self._sock: Optional[socket.socket] = None
        self._listener: Optional[socket.socket] = None
        self._port_file: Optional[str] = None
        if world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        forced = port if port is not None else (int(os.environ["DV_RDZV_PORT"]) if os.environ.get("DV_RDZV_PORT") else None)
        master_port = int(os.environ.get("MASTER_PORT", "29500"))
        via_file = forced is None and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        path = os.path.join(tempfile.gettempdir(), f"dv_rdzv_{os.getuid()}_{os.getppid()}_{master_port}")
        if rank == 0:
            self._serve(addr, 0 if via_file else (forced if forced is not None else master_port), path if via_file else None)
        else:
            self._join(addr, None if via_file else (forced if forced is not None else master_port), path if via_file else None)

    # -- hub ----------------------------------------------------------------------------------------------------------
    def _serve(self, addr: str, port: int, path: Optional[str]):
        token = secrets.token_bytes(16)
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        ls.bind((addr, port))
        ls.listen(self.world)
        ls.settimeout(self.timeout)
        self._listener = ls
        if path is not None:
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as fh:
                fh.write(f"{ls.getsockname()[1]} {token.hex()}\n")
            os.replace(tmp, path)                       # atomic: a reader sees the old file or the new one, never half
            self._port_file = path
        conns: List[Optional[socket.socket]] = [None] * self.world
        deadline = time.monotonic() + self.timeout
        while any(c is None for c in conns[1:]):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {sum(c is None for c in conns[1:])} of {self.world - 1} ranks did not join")
            try:
                c, _ = ls.accept()
            except socket.timeout:
                continue
            try:
                c.settimeout(10.0)
                hello = _recv_exact(c, len(_MAGIC) + 8 + 16)
                r, w = struct.unpack("<II", hello[len(_MAGIC):len(_MAGIC) + 8])
                ok = hello.startswith(_MAGIC) and w == self.world and 0 < r < self.world and conns[r] is None and \
                    (path is None or hello[-16:] == token)
                c.sendall(b"OK" if ok else b"NO")
                if not ok:
                    c.close()
                    continue
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conns[r] = c
            except (OSError, struct.error):             # a port scanner, a client of some other job: not ours
                c.close()
        self._conns = conns

    # -- spoke --------------------------------------------------------------------------------------------------------
    def _join(self, addr: str, port: Optional[int], path: Optional[str]):
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            p, token = port, b"\0" * 16
            if path is not None:
                try:
                    with open(path) as fh:
                        a, b = fh.read().split()
                    p, token = int(a), bytes.fromhex(b)
                except (OSError, ValueError):
                    time.sleep(0.05)
                    continue
            try:
                s = socket.create_connection((addr, p), timeout=5.0)
                s.sendall(_MAGIC + struct.pack("<II", self.rank, self.world) + token)
                if _recv_exact(s, 2) == b"OK":
                    s.settimeout(self.timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s
                    return
                s.close()
                last = "refused by the hub (stale port file or another job)"
            except OSError as e:
                last = repr(e)
            time.sleep(0.1)
        raise TimeoutError(f"rendezvous: rank {self.rank} could not reach rank 0 at {addr} ({last})")

    # -- collectives --------------------------------------------------------------------------------------------------
    def allgather(self, payload: bytes) -> List[bytes]:
        """Every rank contributes a byte string and receives all of them in rank order."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._conns[1:]]
            blob = b"".join(struct.pack("<Q", len(b)) + b for b in parts)
            for c in self._conns[1:]:
                _send_msg(c, blob)
            return parts
        _send_msg(self._sock, payload)
        blob, parts, off = _recv_msg(self._sock), [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def broadcast(self, payload: Optional[bytes], src: int = 0) -> bytes:
        return self.allgather(payload if self.rank == src and payload is not None else b"")[src]

    def barrier(self):
        self.allgather(b"")

    def max(self, value: float) -> float:
        return max(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(value))))

    def gather_object(self, obj, dst: int = 0):
        """Pickled objects of all ranks on `dst` (None elsewhere); the pieces of deblend_sharded(gather=True)."""
        parts = self.allgather(pickle.dumps(obj) if self.rank != dst else b"")
        if self.rank != dst:
            return None
        return [obj if r == dst else pickle.loads(b) for r, b in enumerate(parts)]

    def close(self):
        for c in self._conns:
            if c is not None:
                c.close()
        self._conns = []
        if self._sock is not None:
            self._sock.close()
            self._sock = None
        if self._listener is not None:
            self._listener.close()
            self._listener = None
        if self._port_file:
            try:
                os.unlink(self._port_file)
            except OSError:
                pass
            self._port_file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def exchange_unique_id(rank: int, group) -> bytes:
    """Rank 0 creates the RCCL unique id; every rank returns the same 128 bytes.  `group`: a HostGroup, or any object
    with torch.distributed's broadcast_object_list (kept for callers that already run a process group)."""
    mine = E.Context.unique_id() if rank == 0 else None
    if hasattr(group, "broadcast_object_list"):
        payload = [mine]
        group.broadcast_object_list(payload, src=0)
        return payload[0]
    return group.broadcast(mine, src=0)


def make_context(rank: int, world: int, local_rank: Optional[int] = None, group=None) -> E.Context:
    """The engine context of this rank: GPU `local_rank`, RCCL communicator over `world` ranks.  With world > 1 and no
    `group` the ranks meet through HostGroup(rank, world) (MASTER_ADDR / MASTER_PORT); the group stays attached to the
    context (ctx.group) for the caller's barriers and is closed with it."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if os.environ.get("DV_DEBUG_SAME_GPU"):
        # rehearsal of a multi-rank launch on a one-GPU box: every rank opens device 0 (a real RCCL communicator refuses
        # two ranks on one device; with DV_DEBUG_FAKE_PEERS=1 the engine gives each rank a one-rank communicator instead,
        # so the gradients are NOT summed - the launch line, the rendezvous, the multi-rank event scopes and the step
        # structure run for real, the numbers mean nothing)
        local_rank = 0
    if world == 1:
        ctx = E.Context(local_rank, 0, 1, None)
        ctx.group = group
        return ctx
    own = group is None
    if own:
        group = HostGroup(rank, world)
    uid = exchange_unique_id(rank, group)
    ctx = E.Context(local_rank, rank, world, uid)
    ctx.group = group
    ctx._owns_group = own
    return ctx
