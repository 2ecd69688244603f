"""Data-parallel plumbing: one process per GPU, ranks joined through RCCL inside the engine.

The host side only has to (1) hand rank 0's RCCL unique id to every rank, (2) split a global batch / an inference set
into contiguous per-rank shards and (3) offer the launcher-level barrier / max-over-ranks bench.py
needs.  All of it runs over a plain TCP star (`HostGroup`, rank 0 is the hub) found through the MASTER_ADDR /
MASTER_PORT / RANK / WORLD_SIZE variables every launcher exports: no torch (or any other framework) is imported into a
process that drives a GPU.  The reference is single-process (training/train.py:27-37); this layer is north_star's.
"""
from __future__ import annotations

import hmac
import json
import os
import secrets
import socket
import stat
import struct
import sys
import tempfile
import time
from typing import List, Optional, Tuple

import numpy as np

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
_MAGIC = b"DVRDZV02"
_TOKEN_BYTES = 16


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


# -- payloads: a closed set of plain types, never pickle ------------------------------------------------------------------
# What ranks exchange through gather_object are tuples of ints and float arrays (the pieces of deblend_sharded).  They
# travel as a JSON skeleton plus raw array bytes; decoding builds nothing but None / bool / int / float / str / list /
# tuple / dict / numeric ndarray, so a peer's bytes can never run code on rank 0 (pickle.loads on a socket).
def _pack(obj) -> bytes:
    blobs: List[bytes] = []

    def enc(o):
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        if isinstance(o, (np.integer, np.floating)):
            return o.item()
        if isinstance(o, np.ndarray):
            if o.dtype.kind not in "biuf":
                raise TypeError(f"only numeric arrays travel between ranks, not dtype {o.dtype}")
            a = np.ascontiguousarray(o)
            blobs.append(a.tobytes())
            return {"__nd__": len(blobs) - 1, "dtype": a.dtype.str, "shape": list(a.shape)}
        if isinstance(o, tuple):
            return {"__tuple__": [enc(v) for v in o]}
        if isinstance(o, list):
            return [enc(v) for v in o]
        if isinstance(o, dict):
            if not all(isinstance(k, str) for k in o):
                raise TypeError("only string keys travel between ranks")
            return {"__dict__": {k: enc(v) for k, v in o.items()}}
        raise TypeError(f"objects of type {type(o).__name__} do not travel between ranks (plain types and numeric arrays do)")

    head = json.dumps(enc(obj)).encode()
    out = [struct.pack("<QI", len(head), len(blobs)), head]
    for b in blobs:
        out.append(struct.pack("<Q", len(b)))
        out.append(b)
    return b"".join(out)


def _unpack(buf: bytes):
    hl, nb = struct.unpack_from("<QI", buf, 0)
    off = 12
    skel = json.loads(buf[off:off + hl].decode())
    off += hl
    blobs = []
    for _ in range(nb):
        (n,) = struct.unpack_from("<Q", buf, off)
        blobs.append(memoryview(buf)[off + 8:off + 8 + n])
        off += 8 + n

    def dec(o):
        if isinstance(o, list):
            return [dec(v) for v in o]
        if isinstance(o, dict):
            if "__nd__" in o:
                dt = np.dtype(str(o["dtype"]))
                if dt.kind not in "biuf":
                    raise ValueError("refusing a non-numeric array from a peer")
                shape = tuple(int(v) for v in o["shape"])
                a = np.frombuffer(blobs[int(o["__nd__"])], dtype=dt)
                return a.reshape(shape).copy()
            if "__tuple__" in o:
                return tuple(dec(v) for v in o["__tuple__"])
            if "__dict__" in o:
                return {k: dec(v) for k, v in o["__dict__"].items()}
            raise ValueError("malformed payload from a peer")
        return o

    return dec(skel)


def _private_dir() -> str:
    """A directory only this user can enter (0700, owned by us, not a symlink): where rank 0 leaves the job's token."""
    base = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
    d = os.path.join(base, f"dv_rdzv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"{d} must be a directory owned by uid {os.getuid()} with mode 0700 (rendezvous secrets live there)")
    return d


class HostGroup:
    """The ranks of one job as a TCP star with rank 0 in the middle: all-gather of small byte strings, broadcast /
    barrier / max on top of it, and a gather of plain payloads to one rank.  Meant for a handful of messages per run (RCCL
    id, barriers around a timed region, one float) plus the result pieces of sharded inference; everything that matters
    for speed goes through RCCL inside the engine.

    Where the ranks meet.  Launchers export MASTER_ADDR / MASTER_PORT.  `python -m torch.distributed.run` keeps its own
    store listening on that port for the life of the job (TORCHELASTIC_USE_AGENT_STORE=True), so there rank 0 listens on
    an ephemeral port of MASTER_ADDR; any other launcher (mpirun, srun, a shell loop) leaves MASTER_PORT free and rank 0
    listens on it directly.  DV_RDZV_PORT forces a port.

    Who may join.  EVERY handshake carries a 16-byte secret, in both modes (the direct mode used to admit any
    process that could reach the port).  The secret is DV_RDZV_TOKEN (hex, exported by the launcher) if set; otherwise
    rank 0 draws one and leaves it - with the ephemeral port, if any - in a file the ranks of THIS launch can derive: a
    0700 directory of the user (XDG_RUNTIME_DIR or the temp dir), file name from MASTER_PORT and, under torchrun, the
    launcher pid; created with O_EXCL | O_NOFOLLOW, mode 0600.  A stale file of a dead job is refused by the hub's token
    check and the client reads the file again.  Single node, like the engine (one RCCL communicator over the GPUs of a
    box): ranks on other hosts need DV_RDZV_TOKEN.

    Timeouts.  `timeout` (DV_RDZV_TIMEOUT, default 300 s) bounds the rendezvous; afterwards the collectives wait as long
    as the job takes (sharded inference over a million cutouts is minutes) unless `collective_timeout`
    (DV_COLLECTIVE_TIMEOUT) is given."""

    def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None,
                 timeout: Optional[float] = None, collective_timeout: Optional[float] = None):
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        if timeout is None:
            timeout = float(os.environ.get("DV_RDZV_TIMEOUT", "300"))
        if collective_timeout is None and os.environ.get("DV_COLLECTIVE_TIMEOUT"):
            collective_timeout = float(os.environ["DV_COLLECTIVE_TIMEOUT"])
        self.rank, self.world, self.timeout, self.collective_timeout = rank, world, timeout, collective_timeout
        self._conns: List[Optional[socket.socket]] = []
        self._sock: Optional[socket.socket] = None
        self._listener: Optional[socket.socket] = None
        self._token_file: Optional[str] = None
        if world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        forced = port if port is not None else (int(os.environ["DV_RDZV_PORT"]) if os.environ.get("DV_RDZV_PORT") else None)
        master_port = int(os.environ.get("MASTER_PORT", "29500"))
        ephemeral = forced is None and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        env_token = os.environ.get("DV_RDZV_TOKEN")
        if env_token is not None:
            try:
                env_token_b = bytes.fromhex(env_token)
            except ValueError:
                raise ValueError("DV_RDZV_TOKEN must be hex") from None
            if len(env_token_b) != _TOKEN_BYTES:
                raise ValueError(f"DV_RDZV_TOKEN must be {_TOKEN_BYTES} bytes of hex")
        else:
            env_token_b = None
        # the file is needed whenever something is not known to every rank up front: the secret, the ephemeral port
        path = None
        if env_token_b is None or ephemeral:
            tag = f"{master_port}_{os.getppid()}" if ephemeral else f"{forced if forced is not None else master_port}"
            path = os.path.join(_private_dir(), f"job_{tag}")
        try:
            if rank == 0:
                self._serve(addr, 0 if ephemeral else (forced if forced is not None else master_port), path, env_token_b)
            else:
                self._join(addr, None if ephemeral else (forced if forced is not None else master_port), path, env_token_b)
        except BaseException:
            self.close()                        # listener closed, token file unlinked: nothing is left behind on a timeout
            raise

    # -- hub ----------------------------------------------------------------------------------------------------------
    def _serve(self, addr: str, port: int, path: Optional[str], env_token: Optional[bytes]):
        token = env_token if env_token is not None else secrets.token_bytes(_TOKEN_BYTES)
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._listener = ls
        ls.bind((addr, port))
        ls.listen(self.world)
        ls.settimeout(min(self.timeout, 1.0))
        if path is not None:
            try:
                os.unlink(path)                         # a dead job's file (our directory: nobody else can have put it there)
            except FileNotFoundError:
                pass
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            self._token_file = path
            with os.fdopen(fd, "w") as fh:
                fh.write(f"{ls.getsockname()[1]} {token.hex()}\n")
        conns: List[Optional[socket.socket]] = [None] * self.world
        deadline = time.monotonic() + self.timeout
        while any(c is None for c in conns[1:]):
            if time.monotonic() > deadline:
                for c in conns:
                    if c is not None:
                        c.close()
                raise TimeoutError(f"rendezvous: {sum(c is None for c in conns[1:])} of {self.world - 1} ranks did not join")
            try:
                c, _ = ls.accept()
            except socket.timeout:
                continue
            try:
                c.settimeout(10.0)
                hello = _recv_exact(c, len(_MAGIC) + 8 + _TOKEN_BYTES)
                r, w = struct.unpack("<II", hello[len(_MAGIC):len(_MAGIC) + 8])
                ok = hello.startswith(_MAGIC) and w == self.world and 0 < r < self.world and conns[r] is None and \
                    hmac.compare_digest(hello[-_TOKEN_BYTES:], token)
                c.sendall(b"OK" if ok else b"NO")
                if not ok:
                    c.close()
                    continue
                c.settimeout(self.collective_timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conns[r] = c
            except (OSError, struct.error):             # a port scanner, a client of some other job: not ours
                c.close()
        self._conns = conns

    # -- spoke --------------------------------------------------------------------------------------------------------
    def _join(self, addr: str, port: Optional[int], path: Optional[str], env_token: Optional[bytes]):
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            p, token = port, env_token
            if path is not None:
                try:
                    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                    with os.fdopen(fd) as fh:
                        a, b = fh.read().split()
                    if p is None:
                        p = int(a)
                    if token is None:
                        token = bytes.fromhex(b)
                    if len(token) != _TOKEN_BYTES:
                        raise ValueError("short token")
                except (OSError, ValueError):
                    last = "no token file yet"
                    time.sleep(0.05)
                    continue
            try:
                s = socket.create_connection((addr, p), timeout=5.0)
                s.sendall(_MAGIC + struct.pack("<II", self.rank, self.world) + token)
                if _recv_exact(s, 2) == b"OK":
                    s.settimeout(self.collective_timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s
                    return
                s.close()
                last = "refused by the hub (stale token file or another job)"
            except OSError as e:
                last = repr(e)
            time.sleep(0.1)
        raise TimeoutError(f"rendezvous: rank {self.rank} could not reach rank 0 at {addr} ({last})")

    # -- collectives --------------------------------------------------------------------------------------------------
    def allgather(self, payload: bytes) -> List[bytes]:
        """Every rank contributes a (small) byte string and receives all of them in rank order."""
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

    def gather(self, payload: bytes, dst: int = 0) -> Optional[List[bytes]]:
        """Byte strings of all ranks on `dst` (None elsewhere).  The pieces travel to the hub only - and from there to
        `dst` when that is another rank; nothing is echoed to ranks that would throw it away (the allgather form
        sent world x the full result through rank 0's sockets)."""
        if self.world == 1:
            return [payload]
        if not (0 <= dst < self.world):
            raise ValueError(f"bad destination rank {dst}")
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._conns[1:]]
            if dst == 0:
                for c in self._conns[1:]:
                    c.sendall(b"A")                     # 1-byte ack: a spoke leaves the call once the hub holds its piece
                return parts
            for r, c in enumerate(self._conns[1:], start=1):
                if r == dst:
                    _send_msg(c, b"".join(struct.pack("<Q", len(b)) + b for b in parts))
                else:
                    c.sendall(b"A")
            return None
        _send_msg(self._sock, payload)
        if self.rank != dst:
            _recv_exact(self._sock, 1)
            return None
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
        """Payloads of all ranks on `dst` (None elsewhere); the pieces of deblend_sharded(gather=True).  Plain types and
        numeric arrays only (None, bool, int, float, str, list, tuple, dict with string keys, ndarray): they are encoded
        as JSON + raw bytes, never pickled."""
        parts = self.gather(_pack(obj) if self.rank != dst else b"", dst)
        if parts is None:
            return None
        return [obj if r == dst else _unpack(b) for r, b in enumerate(parts)]

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
        if self._token_file:
            try:
                os.unlink(self._token_file)
            except OSError:
                pass
            self._token_file = None

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


def _rehearsal(var: str) -> bool:
    """The one-GPU rehearsal hooks (DV_DEBUG_SAME_GPU: every rank opens device 0; DV_DEBUG_FAKE_PEERS: every rank gets a
    one-rank communicator, gradients are NOT summed) exist in the DEVELOPMENT build of the engine only (selected with
    DEBVADER_AMD_LIB; dv_build_kind() == 1).  With the product library the variables are not honoured - a stray one in a user's shell must
    not turn a job into a rehearsal - and one line on stderr says so."""
    v = os.environ.get(var)
    if not v or v == "0":
        return False
    from . import _lib as L
    if L.IS_DEBUG_LIB:
        return True
    if var not in _rehearsal.told:
        _rehearsal.told.add(var)
        print(f"[debvader_amd] {var} is set but the product library is loaded: ignored (the rehearsal hooks live in the "
              f"development build of the engine, selected with DEBVADER_AMD_LIB)", file=sys.stderr, flush=True)
    return False


_rehearsal.told = set()


def _agree_on_devices(group, rank: int, world: int, local_rank: int, problem: Optional[str]):
    """Every rank tells the others which physical GPU it is about to open - (host name, PCI bus id) - together with any
    problem it found on its own (LOCAL_RANK beyond the visible GPUs), and ALL ranks raise together when one of them has a
    problem or two of them name the same GPU: e.g. a launcher that starts N ranks with a single visible, un-pinned GPU
    sends every rank to device 0.  Done before the communicator exists, so nobody is left waiting inside
    ncclCommInitRank or in the rendezvous for a rank that has already given up.  Every rank of a job takes part (the
    decision to skip - a rehearsal, a group without a byte all-gather - depends only on things all ranks share: the
    launcher's environment and the group's type)."""
    if not hasattr(group, "allgather") or hasattr(group, "get_backend"):
        if problem:
            raise RuntimeError(problem)
        return
    rehearsal = _rehearsal("DV_DEBUG_SAME_GPU") or _rehearsal("DV_DEBUG_FAKE_PEERS")
    bus = "" if problem else getattr(E, "device_bus_id", lambda d: "")(local_rank)
    mine = f"{socket.gethostname()}|{bus}|{problem or ''}".encode()
    seen = {}
    clash = None
    problems = []
    for r, who in enumerate(group.allgather(mine)):
        host, bus_r, prob_r = bytes(who).decode().split("|", 2)
        if prob_r:
            problems.append(f"rank {r}: {prob_r}")
            continue
        if not bus_r or rehearsal:
            continue                       # that rank could not name its device (no GPU visible) / ranks share on purpose
        key = (host, bus_r)
        if key in seen and clash is None:
            clash = (seen[key], r, key)
        seen.setdefault(key, r)
    if problems:
        raise RuntimeError("; ".join(problems))
    if clash:
        raise RuntimeError(f"ranks {clash[0]} and {clash[1]} of {world} would both open GPU {clash[2][1]} on host "
                           f"{clash[2][0]}: RCCL needs one GPU per rank - check LOCAL_RANK and the per-rank "
                           f"HIP_VISIBLE_DEVICES pinning of the launcher")


def make_context(rank: int, world: int, local_rank: Optional[int] = None, group=None) -> E.Context:
    """The engine context of this rank: GPU `local_rank`, RCCL communicator over `world` ranks.  With world > 1 and no
    `group` the ranks meet through HostGroup(rank, world) (MASTER_ADDR / MASTER_PORT); the group stays attached to the
    context (ctx.group) for the caller's barriers and is closed with it.  Order with world > 1: join the group FIRST, then
    agree on the devices (a rank that cannot open its GPU says so to everybody and all ranks raise the same error - no
    healthy peer is left blocking in the rendezvous), then the communicator; a group created here is closed again when
    any of that fails."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
    same_gpu = _rehearsal("DV_DEBUG_SAME_GPU")
    fake_peers = _rehearsal("DV_DEBUG_FAKE_PEERS")
    for var, on in (("DV_DEBUG_SAME_GPU", same_gpu), ("DV_DEBUG_FAKE_PEERS", fake_peers)):
        if on and world > 1:
            print(f"[debvader_amd] WARNING: {var} is set - this is a REHEARSAL of a {world}-rank launch on one GPU; "
                  f"gradients are NOT summed across ranks and any throughput it prints means nothing", file=sys.stderr,
                  flush=True)
    problem = None
    if same_gpu:
        # rehearsal of a multi-rank launch on a one-GPU box: every rank opens device 0 (a real RCCL communicator refuses
        # two ranks on one device; with DV_DEBUG_FAKE_PEERS=1 the development library gives each rank a one-rank
        # communicator instead, so the gradients are NOT summed - the launch line, the rendezvous, the multi-rank event
        # scopes and the step structure run for real, the numbers mean nothing)
        local_rank = 0
    else:
        # launchers that pin one GPU per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per process) leave every rank
        # with a single visible device, which is then device 0 whatever LOCAL_RANK says
        try:
            visible = E.device_count()
        except Exception:
            visible = 0
        if visible == 1:
            local_rank = 0
        elif 1 < visible <= local_rank:
            # (a modulo here would put two ranks on one GPU; RCCL then fails late, inside ncclCommInitRank, or stalls its peers)
            problem = (f"LOCAL_RANK {local_rank} but only {visible} GPUs are visible to this process: launch at most "
                       f"{visible} ranks per node, or pin one GPU per rank (HIP_VISIBLE_DEVICES)")
    if world == 1:
        if problem:
            raise RuntimeError(problem)
        ctx = E.Context(local_rank, 0, 1, None)
        ctx.group = group
        return ctx
    own = group is None
    if own:
        group = HostGroup(rank, world)
    try:
        _agree_on_devices(group, rank, world, local_rank, problem)
        uid = exchange_unique_id(rank, group)
        ctx = E.Context(local_rank, rank, world, uid)
    except BaseException:
        if own:
            try:
                group.close()
            except Exception:
                pass
        raise
    ctx.group = group
    ctx._owns_group = own
    return ctx
