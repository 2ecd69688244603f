"""Reader for TensorFlow tensor-bundle checkpoints (what `net.load_weights(tf.train.latest_checkpoint(dir))`
consumes in the reference: src/debvader/model/model.py:262-266, training/train.py:145-154).

Format (TensorFlow core/util/tensor_bundle): `<prefix>.index` is a LevelDB-style sorted string table — data
blocks of prefix-compressed (key, value) entries with restart arrays, each block followed by a 5-byte trailer
(compression type, masked crc32c), and a 48-byte footer (metaindex handle, index handle, padding, magic
0xdb4775248b80fb57).  The empty key holds a BundleHeaderProto, every other key a BundleEntryProto
{dtype, shape, shard_id, offset, size, crc32c}; `<prefix>.data-000ss-of-000nn` hold the raw little-endian tensors.
Keras object-based checkpoints name variables
`layer_with_weights-<model>/layer_with_weights-<k>/<attr>/.ATTRIBUTES/VARIABLE_VALUE`.

Pure Python + numpy, no TensorFlow.  Only what the engine needs is implemented: uncompressed blocks, float32 /
int64 tensors, no sliced entries.  A writer is not provided (a loadable Keras checkpoint also needs the serialized
object graph).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9
_NP_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


# ---- crc32c (Castagnoli), table driven ------------------------------------------------------------
def _make_crc_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return np.array(tbl, dtype=np.uint32)


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes) -> int:
    crc = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        crc = int(tbl[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / protobuf wire format ---------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf decoder: field number -> list of raw values (int for varint/fixed, bytes for length-delimited)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.setdefault(field, []).append(v)
    return out


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for d in _parse_proto(buf).get(2, []):        # TensorShapeProto.dim
        size = _parse_proto(d).get(1, [0])[0]     # Dim.size (int64 varint)
        dims.append(size if size < (1 << 63) else size - (1 << 64))
    return tuple(dims)


# ---- table reading ------------------------------------------------------------------------------------
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed index blocks (snappy) are not supported")
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if stored != masked_crc32c(data[offset:offset + size + 1]):
            raise ValueError("index block checksum mismatch")
    return block


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


class BundleEntry:
    __slots__ = ("dtype", "shape", "shard_id", "offset", "size", "crc32c")

    def __init__(self, dtype, shape, shard_id, offset, size, crc):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c = dtype, shape, shard_id, offset, size, crc

    def __repr__(self):
        return f"BundleEntry(dtype={self.dtype}, shape={self.shape}, shard={self.shard_id}, offset={self.offset}, size={self.size})"


class TensorBundle:
    """Index of a checkpoint prefix; tensors are read lazily from the data shards."""

    def __init__(self, prefix: str, verify_index: bool = True):
        self.prefix = prefix
        with open(prefix + ".index", "rb") as f:
            data = f.read()
        if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
            raise ValueError(f"{prefix}.index is not a tensor-bundle index (bad magic)")
        footer = data[-48:]
        pos = 0
        _, pos = _varint(footer, pos)      # metaindex handle (offset, size): unused
        _, pos = _varint(footer, pos)
        idx_off, pos = _varint(footer, pos)
        idx_size, pos = _varint(footer, pos)
        self.entries: Dict[str, BundleEntry] = {}
        self.num_shards = 1
        for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify_index)):
            boff, p2 = _varint(handle, 0)
            bsize, _ = _varint(handle, p2)
            for key, value in _block_entries(_read_block(data, boff, bsize, verify_index)):
                msg = _parse_proto(value)
                if key == b"":
                    self.num_shards = msg.get(1, [1])[0]
                    if msg.get(2, [0])[0] != 0:
                        raise NotImplementedError("big-endian bundles are not supported")
                    continue
                if 7 in msg:
                    raise NotImplementedError(f"sliced tensor {key!r} is not supported")
                self.entries[key.decode()] = BundleEntry(
                    msg.get(1, [0])[0], _parse_shape(msg[2][0]) if 2 in msg else (), msg.get(3, [0])[0],
                    msg.get(4, [0])[0], msg.get(5, [0])[0], msg.get(6, [0])[0])

    def keys(self) -> List[str]:
        return sorted(self.entries)

    def shard_path(self, shard_id: int) -> str:
        return f"{self.prefix}.data-{shard_id:05d}-of-{self.num_shards:05d}"

    def read(self, key: str, verify="auto") -> np.ndarray:
        """verify: True / False / "auto" (checksum tensors up to 256 KiB: the pure-Python crc32c costs ~1 s per MiB)."""
        e = self.entries[key]
        if e.dtype not in _NP_DTYPES:
            raise NotImplementedError(f"dtype {e.dtype} of {key} is not supported")
        path = self.shard_path(e.shard_id)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: {key} lives in shard {e.shard_id} of {self.num_shards}")
        with open(path, "rb") as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError(f"{path} is truncated ({key})")
        if verify == "auto":
            verify = e.size <= (256 << 10)
        if verify and e.crc32c and masked_crc32c(raw) != e.crc32c:
            raise ValueError(f"checksum mismatch for {key}")
        return np.frombuffer(raw, dtype=_NP_DTYPES[e.dtype]).reshape(e.shape).copy()


# ---- Keras object-checkpoint names -> engine tensor names ---------------------------------------------------
_ATTR = "/.ATTRIBUTES/VARIABLE_VALUE"


def variable_keys(specs) -> Dict[str, str]:
    """Engine tensor name -> checkpoint key, from the engine's spec list in checkpoint order
    (encoder = layer_with_weights-0, decoder = layer_with_weights-1)."""
    out, counters = {}, {"enc": -1, "dec": -1}
    last_layer = {"enc": None, "dec": None}
    for name, _, _ in specs:
        side, layer, attr = name.split("/")
        if layer != last_layer[side]:
            counters[side] += 1
            last_layer[side] = layer
        model = 0 if side == "enc" else 1
        out[name] = f"layer_with_weights-{model}/layer_with_weights-{counters[side]}/{attr}{_ATTR}"
    return out


def slot_key(var_key: str, slot: str) -> str:
    """Key of an optimizer slot ('m' or 'v') of a variable in a Keras optimizer_v2 checkpoint."""
    return var_key[:-len(_ATTR)] + f"/.OPTIMIZER_SLOT/optimizer/{slot}{_ATTR}"


def latest_checkpoint_prefix(directory: str):
    """tf.train.latest_checkpoint: reads `<dir>/checkpoint` and returns the prefix it names (or None)."""
    f = os.path.join(directory, "checkpoint")
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        for line in fh:
            if line.startswith("model_checkpoint_path:"):
                return os.path.join(directory, line.split(":", 1)[1].strip().strip('"'))
    return None


def load_into_engine(engine, prefix: str, load_slots: bool = False) -> int:
    """Copies every model variable of the bundle into the engine; returns the number of tensors loaded.
    Raises if a variable is missing or has the wrong shape (a wrong architecture must not load silently)."""
    bundle = TensorBundle(prefix)
    keys = variable_keys(engine.specs)
    n = 0
    for i, (name, shape, trainable) in enumerate(engine.specs):
        k = keys[name]
        if k not in bundle.entries:
            raise KeyError(f"{k} ({name}) not found in {prefix}.index")
        if tuple(bundle.entries[k].shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {bundle.entries[k].shape} != model shape {shape}")
        engine.set_param(i, bundle.read(k))
        n += 1
        if load_slots and trainable:
            for which, slot in enumerate(("m", "v")):
                sk = slot_key(k, slot)
                if sk in bundle.entries:
                    engine.set_slot(i, which, bundle.read(sk))
    if load_slots and "optimizer/iter" + _ATTR in bundle.entries:
        engine.iterations = int(np.asarray(bundle.read("optimizer/iter" + _ATTR)).reshape(-1)[0])
    return n
