"""Test-only writer of minimal TensorFlow tensor-bundles (variables only, one uncompressed data block, no object
graph): enough to exercise debvader_amd.model.tf_checkpoint against files laid out like the reference's."""
import struct

import numpy as np

from debvader_amd.model.tf_checkpoint import TABLE_MAGIC, masked_crc32c


def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return _vi((num << 3) | wt) + payload


def _entry_proto(dtype, shape, shard, offset, size, crc):
    dims = b"".join(_field(2, 2, _vi(len(d)) + d) for d in (_field(1, 0, _vi(s)) for s in shape))
    return (_field(1, 0, _vi(dtype)) + _field(2, 2, _vi(len(dims)) + dims) + _field(3, 0, _vi(shard)) +
            _field(4, 0, _vi(offset)) + _field(5, 0, _vi(size)) + _field(6, 5, struct.pack("<I", crc)))


def _block(entries):
    body, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % 16 == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        body += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_bundle(prefix, tensors, num_shards=1, crc_limit=1 << 16):
    """tensors: dict key -> ndarray (float32 / int64).  Tensors above crc_limit bytes get crc32c = 0 (= unchecked),
    because the pure-Python checksum is slow."""
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}
    data, entries = bytearray(), [(b"", _field(1, 0, _vi(num_shards)) + _field(3, 2, _vi(2) + _field(1, 0, _vi(1))))]
    for k in sorted(tensors):
        a = np.ascontiguousarray(tensors[k])
        raw = a.tobytes()
        crc = masked_crc32c(raw) if len(raw) <= crc_limit else 0
        entries.append((k.encode(), _entry_proto(dt[a.dtype], a.shape, 0, len(data), len(raw), crc)))
        data += raw
    with open(f"{prefix}.data-00000-of-{num_shards:05d}", "wb") as f:
        f.write(data)
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00")
        out.extend(struct.pack("<I", masked_crc32c(block + b"\x00")))
        return off, len(block)

    d_off, d_size = emit(_block(entries))
    m_off, m_size = emit(_block([]))
    i_off, i_size = emit(_block([(entries[-1][0] + b"\xff", _vi(d_off) + _vi(d_size))]))
    footer = _vi(m_off) + _vi(m_size) + _vi(i_off) + _vi(i_size)
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(out)
