"""Reader for TensorFlow checkpoints (TensorBundle: `<prefix>.index` + `<prefix>.data-00000-of-00001`) without
TensorFlow — what `tf.train.Checkpoint(net=model).restore(ckpt)` consumes in the reference (util/io.py:36-45,
trainvali.py:134-141), so the authors' released NeRF / NeRFactor weights can be loaded into these models.

Formats (restated from TensorFlow's public sources; no file of the reference is involved):
  * `.index` is an SSTable in the LevelDB table format (tensorflow/core/lib/io/table): data blocks of prefix-compressed
    (shared, unshared, value_len, key_delta, value) entries + restart array, an index block of BlockHandles, a 48-byte
    footer ending in the magic 0xdb4775248b80fb57; every block is followed by a 5-byte trailer (compression type,
    masked crc32c).  The bundle writer uses no compression.
  * key "" -> BundleHeaderProto; every other key -> BundleEntryProto {1 dtype, 2 shape {2 dim {1 size}}, 3 shard_id,
    4 offset, 5 size, 6 crc32c}; tensor bytes are little-endian row-major in the data shard.
  * object-based (TF2) checkpoints name variables `<attr path>/.ATTRIBUTES/VARIABLE_VALUE`; the reference's models sit
    under `net/`, its layers are attributes `net_<network>_layer<i>` (models/base.py:81-104) with Keras variables
    `kernel` / `bias`, the light probe is `_light`, the BRDF codes `latent_code/_z`.
PARITY UNPINNED: no TensorFlow and no released checkpoint is available in this environment; the reader is tested
against an independent writer of the same public format (tests/tf_bundle_writer.py) only."""
import glob
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          14: None, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}   # 14 = bfloat16 (handled below)
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(data, offset, size):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed SSTable block (type %d): the bundle writer emits none" % ctype)
    return raw


def _block_entries(block):
    """Yields (key, value) of one table block (prefix-compressed entries, restart array at the end)."""
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path):
    """All (key, value) pairs of an SSTable file, in key order."""
    data = open(path, 'rb').read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != MAGIC:
        raise ValueError("%s is not an SSTable (bad magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)    # index handle
    idx_size, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_block(data, idx_off, idx_size)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        out.extend(_block_entries(_block(data, off, size)))
    return out


def _parse_message(buf):
    """Minimal protobuf wire decoder: {field: [values]} with varints as ints, fixed32 as ints, bytes as bytes."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        out.setdefault(field, []).append(val)
    return out


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def parse_entry(value):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size)."""
    msg = _parse_message(value)
    shape = []
    if 2 in msg:
        for dim in _parse_message(msg[2][0]).get(2, []):
            shape.append(_signed(_parse_message(dim).get(1, [0])[0]))
    if 7 in msg:
        raise NotImplementedError("sliced (partitioned) variables are not supported")
    return {'dtype': msg.get(1, [0])[0], 'shape': tuple(shape), 'shard_id': msg.get(3, [0])[0],
            'offset': msg.get(4, [0])[0], 'size': msg.get(5, [0])[0]}


def read_index(prefix):
    """{tensor name: entry dict} and the header of a checkpoint prefix (path without '.index')."""
    entries, header = {}, None
    for key, value in read_table(prefix + '.index'):
        if key == b'':
            header = _parse_message(value)   # {1 num_shards, 2 endianness, 3 version}
            if header.get(2, [0])[0] != 0:
                raise NotImplementedError("big-endian bundle")
        else:
            entries[key.decode()] = parse_entry(value)
    return entries, header


def load_tensors(prefix, names=None):
    """{name: np.ndarray} for the numeric tensors of the checkpoint (strings, e.g. the object graph, are skipped)."""
    entries, header = read_index(prefix)
    n_shards = header.get(1, [1])[0] if header else 1
    shards = {}
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['dtype'] not in DTYPES:
            continue                                   # DT_STRING and friends
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, n_shards), dtype=np.uint8, mode='r')
        raw = np.asarray(shards[sid][e['offset']:e['offset'] + e['size']])
        if e['dtype'] == 14:                           # bfloat16 -> float32
            arr = (raw.view('<u2').astype(np.uint32) << 16).view(np.float32)
        else:
            arr = raw.view(np.dtype(DTYPES[e['dtype']]).newbyteorder('<'))
        out[name] = np.array(arr).reshape(e['shape'])
    return out


def is_tf_checkpoint(path):
    return bool(glob.glob(glob.escape(path) + '.index'))


def to_state_dict(tensors, root='net'):
    """TF object-graph variable names under `root` -> names of a torch state_dict of the mirrored model:
    'net/net_coarse_enc_layer0/kernel/.ATTRIBUTES/VARIABLE_VALUE' -> 'net_coarse_enc_layer0.kernel'."""
    pref = root + '/'
    out = {}
    for name, arr in tensors.items():
        if not name.startswith(pref) or not name.endswith(SUFFIX):
            continue
        out[name[len(pref):-len(SUFFIX)].replace('/', '.')] = arr
    return out
