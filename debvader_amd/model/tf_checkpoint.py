"""Reader for TensorFlow tensor-bundle checkpoints (what `net.load_weights(tf.train.latest_checkpoint(dir))`
consumes in the reference: src/debvader/model/model.py:262-266, training/train.py:145-154).

Format (TensorFlow core/util/tensor_bundle): `<prefix>.index` is a LevelDB-style sorted string table — data
blocks of prefix-compressed (key, value) entries with restart arrays, each block followed by a 5-byte trailer
(compression type, masked crc32c), and a 48-byte footer (metaindex handle, index handle, padding, magic
0xdb4775248b80fb57).  The empty key holds a BundleHeaderProto, every other key a BundleEntryProto
{dtype, shape, shard_id, offset, size, crc32c}; `<prefix>.data-000ss-of-000nn` hold the raw little-endian tensors.
Keras object-based checkpoints name variables
`layer_with_weights-<model>/layer_with_weights-<k>/<attr>/.ATTRIBUTES/VARIABLE_VALUE`.

Pure Python + numpy, no TensorFlow (checksums through the engine library's host-side dv_crc32c when it is loadable).
Only what the engine needs is implemented: uncompressed blocks, float32 / int64 tensors, no sliced entries.

The writer (`save_from_engine`) produces what `ModelCheckpoint(save_weights_only=True)` leaves behind in the
reference (train.py:49-75): `<prefix>.index`, `<prefix>.data-00000-of-00001` and the `checkpoint` state file, with
the `_CHECKPOINTABLE_OBJECT_GRAPH` entry (a TrackableObjectGraph proto: root -> encoder / decoder Functional models
-> layers -> variables, optimizer -> hyper-parameters and m / v slots) that Keras `load_weights` walks.  Pinned by
the reference's own checkpoint: re-encoding its parsed index reproduces `weights_noisy_v4...ckpt.index` byte for
byte, the string-tensor checksum recipe reproduces its stored crc, and the generated object graph has the same
variable paths, keys, names and slot list as the one in its data shard 0 (tests/test_tf_checkpoint.py).
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


def _native_crc():
    try:
        from debvader_amd._lib import lib
        return lib.dv_crc32c
    except Exception:       # library not built: fall back to the table loop (slow, checkpoints only)
        return None


_NATIVE_CRC = _native_crc()


def crc32c_extend(crc: int, data: bytes) -> int:
    """tensorflow::crc32c::Extend: checksum of the concatenation given the checksum of the prefix (0 to start)."""
    if _NATIVE_CRC is not None and len(data) >= 64:
        return int(_NATIVE_CRC(crc, bytes(data), len(data)))
    crc ^= 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        crc = int(tbl[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def crc32c(data: bytes) -> int:
    return crc32c_extend(0, data)


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


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

    def read_string(self, key: str) -> bytes:
        """A scalar DT_STRING entry ([varint length][masked crc of the length][bytes]), e.g. the object graph."""
        e = self.entries[key]
        if e.dtype != DT_STRING or e.shape != ():
            raise ValueError(f"{key} is not a scalar string")
        with open(self.shard_path(e.shard_id), "rb") as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        n, pos = _varint(raw, 0)
        body = raw[pos + 4:pos + 4 + n]
        if len(body) != n:
            raise ValueError(f"{key} is truncated")
        if e.crc32c and string_tensor_crc(body) != e.crc32c:
            raise ValueError(f"checksum mismatch for {key}")
        return body

    def read(self, key: str, verify="auto") -> np.ndarray:
        """verify: True / False / "auto" (always with the native checksum; up to 256 KiB with the Python loop)."""
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
            verify = _NATIVE_CRC is not None or e.size <= (256 << 10)
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


# ---- writer -------------------------------------------------------------------------------------------------
def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb(num: int, wt: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | wt) + payload


def _pb_bytes(num: int, b: bytes) -> bytes:
    return _pb(num, 2, _enc_varint(len(b)) + b)


def string_tensor_crc(body: bytes) -> int:
    """Entry checksum of a scalar string tensor: lengths (as uint32), the masked length checksum, then the bytes
    (TensorFlow tensor_bundle WriteStringTensor; reproduces the crc stored for the reference's object graph)."""
    c = crc32c_extend(0, struct.pack("<I", len(body)))
    c = crc32c_extend(c, struct.pack("<I", mask_crc(c)))
    return mask_crc(crc32c_extend(c, body))


def _string_tensor_bytes(body: bytes) -> bytes:
    return _enc_varint(len(body)) + struct.pack("<I", mask_crc(crc32c_extend(0, struct.pack("<I", len(body))))) + body


def _entry_bytes(e: "BundleEntry") -> bytes:
    """BundleEntryProto, proto3 style (zero-valued scalar fields omitted, the shape message always present)."""
    dims = b"".join(_pb_bytes(2, _pb(1, 0, _enc_varint(d))) for d in e.shape)
    out = _pb(1, 0, _enc_varint(e.dtype)) + _pb_bytes(2, dims)
    if e.shard_id:
        out += _pb(3, 0, _enc_varint(e.shard_id))
    if e.offset:
        out += _pb(4, 0, _enc_varint(e.offset))
    if e.size:
        out += _pb(5, 0, _enc_varint(e.size))
    if e.crc32c:
        out += _pb(6, 5, struct.pack("<I", e.crc32c))
    return out


def _table_block(items) -> bytes:
    """LevelDB block: prefix-compressed entries, a restart point every 16 entries."""
    body, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % 16 == 0:
            restarts.append(len(body))
        else:
            lim = min(len(prev), len(k))
            while shared < lim and prev[shared] == k[shared]:
                shared += 1
        body += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _short_successor(key: bytes) -> bytes:
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def encode_index(entries, num_shards: int = 1, block_size: int = 262144) -> bytes:
    """Bytes of `<prefix>.index` for `entries` = {key: BundleEntry}: header entry, sorted keys, uncompressed blocks
    (table block_size 256 KiB like TensorFlow's writer), empty metaindex, index block keyed by short separators."""
    header = _pb(1, 0, _enc_varint(num_shards)) + _pb_bytes(3, _pb(1, 0, _enc_varint(1)))
    items = [(b"", header)] + [(k.encode(), _entry_bytes(entries[k])) for k in sorted(entries, key=lambda s: s.encode())]
    out, index_items = bytearray(), []
    start = 0
    while start < len(items):
        end, size = start, 0
        while end < len(items) and (end == start or size < block_size):
            size += len(items[end][0]) + len(items[end][1]) + 3
            end += 1
        block = _table_block(items[start:end])
        handle = _enc_varint(len(out)) + _enc_varint(len(block))
        out += block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))
        last = items[end - 1][0]
        if end < len(items):      # shortest separator between this block's last key and the next block's first
            nxt, i = items[end][0], 0
            while i < min(len(last), len(nxt)) and last[i] == nxt[i]:
                i += 1
            sep = last[:i] + bytes([last[i] + 1]) if i < len(last) and last[i] + 1 < nxt[i] else last
        else:
            sep = _short_successor(last)
        index_items.append((sep, handle))
        start = end
    meta = _table_block([])
    meta_handle = _enc_varint(len(out)) + _enc_varint(len(meta))
    out += meta + b"\x00" + struct.pack("<I", mask_crc(crc32c(meta + b"\x00")))
    index = _table_block(index_items)
    index_handle = _enc_varint(len(out)) + _enc_varint(len(index))
    out += index + b"\x00" + struct.pack("<I", mask_crc(crc32c(index + b"\x00")))
    footer = meta_handle + index_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    return bytes(out)


_DT_OF = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int64): DT_INT64, np.dtype(np.float64): 2, np.dtype(np.int32): 3}


def atomic_write(path: str, payload: bytes) -> None:
    tmp = f"{path}.tmp-{os.getpid()}"
    try:
        with open(tmp, "wb") as f:
            f.write(payload)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def write_bundle(prefix: str, tensors) -> None:
    """Writes `<prefix>.index` and `<prefix>.data-00000-of-00001`.  tensors: {key: ndarray or bytes (scalar string)}.
    Tensors are laid out in key order, every entry carries its masked crc32c (TensorFlow verifies them on read)."""
    entries, data = {}, bytearray()
    for k in sorted(tensors, key=lambda s: s.encode()):
        v = tensors[k]
        if isinstance(v, (bytes, bytearray)):
            raw = _string_tensor_bytes(bytes(v))
            entries[k] = BundleEntry(DT_STRING, (), 0, len(data), len(raw), string_tensor_crc(bytes(v)))
        else:
            a = np.asarray(v)                 # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_OF:
                raise TypeError(f"{k}: dtype {a.dtype} cannot be stored")
            raw = a.tobytes(order="C")
            entries[k] = BundleEntry(_DT_OF[a.dtype], tuple(int(d) for d in a.shape), 0, len(data), len(raw),
                                     mask_crc(crc32c(raw)))
        data += raw
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    # data shard first, index next (write_checkpoint_state comes last): each file is written under a temporary name,
    # flushed to disk and renamed, so a crash mid-save leaves the previous checkpoint - ModelCheckpoint(save_best_only)
    # keeps overwriting ONE prefix - intact, as TensorFlow's temp-prefix-then-rename does
    atomic_write(f"{prefix}.data-00000-of-00001", bytes(data))
    atomic_write(prefix + ".index", encode_index(entries, 1))


def write_checkpoint_state(prefix: str) -> None:
    """The `checkpoint` file tf.train.latest_checkpoint reads (relative path, as Keras' ModelCheckpoint writes)."""
    name = os.path.basename(prefix)
    atomic_write(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"),
                 f'model_checkpoint_path: "{name}"\nall_model_checkpoint_paths: "{name}"\n'.encode())


# ---- Keras object graph -----------------------------------------------------------------------------------------
def _keras_layers(specs):
    """Per side, the Keras layer list of the reference models (model.py:61-100, 103-161) as
    (kind, engine layer name or None): weighted layers come from the engine's spec list, the weight-less ones
    (InputLayer, Flatten, Reshape, Cropping2D, DistributionLambda) are placed where the reference builds them."""
    by_side = {"enc": [], "dec": []}
    for name, shape, _ in specs:
        side, layer, attr = name.split("/")
        if not by_side[side] or by_side[side][-1][0] != layer:
            by_side[side].append((layer, [], None))
        by_side[side][-1][1].append(attr)
        if attr in ("kernel", "alpha"):
            by_side[side][-1] = (layer, by_side[side][-1][1], tuple(shape))

    def kind(layer, attrs, side):
        if "gamma" in attrs:
            return "batch_normalization"
        if "alpha" in attrs:
            return "p_re_lu"
        if layer.startswith("convt"):
            return "conv2d_transpose"
        if layer.startswith("conv") or layer == "head":
            return "conv2d"
        return "dense"

    enc = [("input", None, None)]
    for layer, attrs, shape in by_side["enc"]:
        k = kind(layer, attrs, "enc")
        if k == "p_re_lu" and shape is not None and len(shape) == 1 and not any(e[0] == "flatten" for e in enc):
            enc.append(("flatten", None, None))            # model.py:94 Flatten before the 1-D PReLU
        enc.append((k, layer, attrs))
    dec, seen_conv = [("input", None, None)], False
    for layer, attrs, shape in by_side["dec"]:
        k = kind(layer, attrs, "dec")
        if k == "conv2d_transpose" and not seen_conv:
            dec.append(("reshape", None, None))            # model.py:118-119 Reshape before the first Conv2DTranspose
            seen_conv = True
        dec.append((k, layer, attrs))
    return enc, dec


def object_graph(specs, cropping: bool = True, with_optimizer: bool = True) -> bytes:
    """Serialized TrackableObjectGraph for the reference's `net` (Model(inputs, decoder(latent(encoder(x))))) with a
    legacy Adam optimizer.  Node ids are assigned breadth first like TensorFlow's ObjectGraphView; the TFP-internal
    bookkeeping nodes of the two distribution layers (no variables) are left out."""
    enc, dec = _keras_layers(specs)
    if cropping:
        dec.append(("cropping2d", None, None))              # model.py:140-148
    dec.append(("distribution_lambda", None, None))          # model.py:154-159
    keys = variable_keys(specs)
    trainable = {n: t for n, _, t in specs}

    nodes = [{"children": [], "attr": None, "slots": []}]    # node 0 = root

    def new_node():
        nodes.append({"children": [], "attr": None, "slots": []})
        return len(nodes) - 1

    def add_layers(parent, layers):
        """children `layer_with_weights-j` / `layer-i` of a Functional model; returns [(node, entry)]."""
        out, j = [], 0
        for i, entry in enumerate(layers):
            n = new_node()
            if entry[1] is not None:
                nodes[parent]["children"].append((n, f"layer_with_weights-{j}"))
                j += 1
            nodes[parent]["children"].append((n, f"layer-{i}"))
            out.append((n, entry))
        return out

    # breadth first: root's children, then each child's children, ...
    top = add_layers(0, [("input", None, None), ("model", "enc", None), ("distribution_lambda", None, None),
                         ("model", "dec", None)])
    opt = None
    if with_optimizer:
        opt = new_node()
        nodes[0]["children"].append((opt, "optimizer"))
    enc_nodes = add_layers(top[1][0], enc)
    dec_nodes = add_layers(top[3][0], dec)
    if with_optimizer:
        for hp in ("iter", "beta_1", "beta_2", "decay", "learning_rate"):
            n = new_node()
            nodes[opt]["children"].append((n, hp))
            nodes[n]["attr"] = (f"training/Adam/{hp}", f"optimizer/{hp}{_ATTR}")
    counters, var_node, order = {}, {}, []
    for side, lst in (("enc", enc_nodes), ("dec", dec_nodes)):
        for n, (kind, layer, attrs) in lst:
            if layer is None:
                continue
            c = counters.get(kind, 0)
            counters[kind] = c + 1
            kname = kind if c == 0 else f"{kind}_{c}"
            if kind == "batch_normalization":
                nodes[n]["children"].append((new_node(), "axis"))
            for attr in attrs:
                v = new_node()
                nodes[n]["children"].append((v, attr))
                full = f"{side}/{layer}/{attr}"
                nodes[v]["attr"] = (f"{kname}/{attr}", keys[full])
                var_node[full] = v
                if trainable[full]:
                    order.append((full, f"{kname}/{attr}"))
    if with_optimizer:
        for slot in ("m", "v"):
            for full, kname in order:
                v = new_node()
                nodes[v]["attr"] = (f"{kname}/{slot}", slot_key(keys[full], slot))
                nodes[opt]["slots"].append((var_node[full], slot, v))

    out = bytearray()
    for nd in nodes:
        body = b"".join(_pb_bytes(1, (_pb(1, 0, _enc_varint(cid)) if cid else b"") + _pb_bytes(2, name.encode()))
                        for cid, name in nd["children"])
        if nd["attr"]:
            full_name, key = nd["attr"]
            body += _pb_bytes(2, _pb_bytes(1, b"VARIABLE_VALUE") + _pb_bytes(2, full_name.encode()) +
                              _pb_bytes(3, key.encode()))
        for orig, slot, sv in nd["slots"]:
            body += _pb_bytes(3, _pb(1, 0, _enc_varint(orig)) + _pb_bytes(2, slot.encode()) + _pb(3, 0, _enc_varint(sv)))
        out += _pb_bytes(1, body)
    return bytes(out)


def parse_object_graph(blob: bytes):
    """{path tuple: (full_name, checkpoint_key)} for every variable reachable from the root, and the optimizer's
    slot list [(variable key, slot name, slot full_name, slot key)] - the part of a TrackableObjectGraph that decides
    what `load_weights` restores."""
    raw = [_parse_proto(b) for b in _parse_proto(blob).get(1, [])]
    nodes = []
    for n in raw:
        ch = [(_parse_proto(c).get(1, [0])[0], _parse_proto(c).get(2, [b""])[0].decode()) for c in n.get(1, [])]
        at = None
        for a in n.get(2, []):
            p = _parse_proto(a)
            at = (p.get(2, [b""])[0].decode(), p.get(3, [b""])[0].decode())
        sl = [(_parse_proto(a).get(1, [0])[0], _parse_proto(a).get(2, [b""])[0].decode(), _parse_proto(a).get(3, [0])[0])
              for a in n.get(3, [])]
        nodes.append((ch, at, sl))
    paths, seen, queue = {}, {0}, [((), 0)]
    while queue:
        path, i = queue.pop(0)
        ch, at, _ = nodes[i]
        if at:
            paths[path] = at
        for cid, name in ch:
            if at is None and nodes[cid][1] is not None:
                paths[path + (name,)] = nodes[cid][1]          # every alias path of a variable is recorded
            if cid not in seen:
                seen.add(cid)
                queue.append((path + (name,), cid))
    slots = []
    for ch, at, sl in nodes:
        for orig, slot, sv in sl:
            slots.append((nodes[orig][1][1], slot, nodes[sv][1][0], nodes[sv][1][1]))
    return paths, slots


def save_from_engine(engine, prefix: str, optimizer=None, cropping=None, write_state: bool = True) -> None:
    """Saves the engine's variables (and, with `optimizer` = dict(learning_rate, beta_1, beta_2[, decay]), the Adam
    hyper-parameters, step counter and m / v slots) as a TensorFlow checkpoint `prefix` that the reference's
    `net.load_weights(tf.train.latest_checkpoint(dir))` (model.py:262-266, train.py:145-154) can restore."""
    keys = variable_keys(engine.specs)
    tensors = {}
    for i, (name, shape, trainable) in enumerate(engine.specs):
        tensors[keys[name]] = np.asarray(engine.get_param(i), dtype=np.float32).reshape(shape)
        if optimizer is not None and trainable:
            for which, slot in enumerate(("m", "v")):
                tensors[slot_key(keys[name], slot)] = np.asarray(engine.get_slot(i, which), np.float32).reshape(shape)
    if optimizer is not None:
        tensors["optimizer/iter" + _ATTR] = np.array(int(engine.iterations), dtype=np.int64)
        for hp in ("beta_1", "beta_2", "decay", "learning_rate"):
            tensors[f"optimizer/{hp}{_ATTR}"] = np.array(float(optimizer.get(hp, 0.0)), dtype=np.float32)
    if cropping is None:
        cfg = engine.cfg
        full = -(-cfg.height // (1 << cfg.n_levels)) * (1 << cfg.n_levels)
        cropping = full != cfg.height
    tensors["_CHECKPOINTABLE_OBJECT_GRAPH"] = object_graph(engine.specs, cropping, optimizer is not None)
    write_bundle(prefix, tensors)
    if write_state:
        write_checkpoint_state(prefix)
