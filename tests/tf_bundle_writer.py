"""Independent WRITER of the TensorBundle checkpoint format (test infrastructure for util/tf_ckpt.py): an SSTable
builder following LevelDB's table_format.md (prefix-compressed entries, restart points every 16 keys, 4 KiB blocks,
index block, 48-byte footer, per-block trailer with masked crc32c) and a protobuf encoder for BundleHeaderProto /
BundleEntryProto."""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
DT_STRING = 7


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82f63b78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xffffffff
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


class BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + \
            struct.pack('<I', len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: sorted list of (key bytes, value bytes)."""
    out = bytearray()
    index = BlockBuilder(restart_interval=1)

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes)
        out.extend(b'\x00' + struct.pack('<I', masked_crc(block_bytes + b'\x00')))
        return off, len(block_bytes)
    bb, last_key = BlockBuilder(), None
    for key, value in items:
        bb.add(key, value)
        last_key = key
        if len(bb.buf) >= block_size:
            off, size = emit(bb.finish())
            index.add(last_key, varint(off) + varint(size))
            bb = BlockBuilder()
    if bb.count:
        off, size = emit(bb.finish())
        index.add(last_key, varint(off) + varint(size))
    meta_off, meta_size = emit(BlockBuilder().finish())          # empty metaindex block
    idx_off, idx_size = emit(index.finish())
    footer = varint(meta_off) + varint(meta_size) + varint(idx_off) + varint(idx_size)
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC))
    open(path, 'wb').write(bytes(out))


def _field(num, wire, payload):
    return varint((num << 3) | wire) + payload


def entry_proto(dtype, shape, offset, size, crc):
    dims = b''.join(_field(2, 2, varint(len(d)) + d) for d in (_field(1, 0, varint(s)) for s in shape))
    msg = _field(1, 0, varint(dtype))
    msg += _field(2, 2, varint(len(dims)) + dims)
    if offset:
        msg += _field(4, 0, varint(offset))
    msg += _field(5, 0, varint(size)) + _field(6, 5, struct.pack('<I', crc))
    return msg


def write_bundle(prefix, tensors, strings=None):
    """tensors: {name: np.ndarray}; strings: {name: bytes} stored as scalar DT_STRING entries (skipped by the reader)."""
    items, data = [], bytearray()
    header = _field(1, 0, varint(1)) + _field(3, 2, varint(2) + _field(1, 0, varint(1)))   # num_shards=1, version{1}
    items.append((b'', header))
    names = sorted(list(tensors) + list(strings or {}))
    for name in names:
        if strings and name in strings:
            payload = varint(len(strings[name])) + strings[name]     # length-prefixed string element
            items.append((name.encode(), entry_proto(DT_STRING, (), len(data), len(payload), masked_crc(payload))))
            data += payload
            continue
        arr = np.asarray(tensors[name])   # (ascontiguousarray would turn a scalar into shape (1,))
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        items.append((name.encode(), entry_proto(DT[arr.dtype], arr.shape, len(data), len(raw), 0)))
        data += raw
    items.sort(key=lambda kv: kv[0])
    write_table(prefix + '.index', items)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
