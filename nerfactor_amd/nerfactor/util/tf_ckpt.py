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
  * the string entry `_CHECKPOINTABLE_OBJECT_GRAPH` holds a TrackableObjectGraph proto: nodes {1 children {1 node_id,
    2 local_name}, 2 attributes {1 name, 2 full_name, 3 checkpoint_key}}.  A variable's checkpoint key is the FIRST path
    the saver found to it; for tf.keras.Model subclasses the automatic `layer_with_weights-N` dependencies come before
    attribute names, so released checkpoints may be keyed `net/layer_with_weights-0/kernel/...`.  `to_state_dict`
    therefore resolves every model attribute path (`net/net_coarse_enc_layer0/kernel`) by WALKING THE OBJECT GRAPH to
    the node and taking whatever key that node was saved under; plain key matching is the fallback for bundles
    without a graph.
PARITY: no TensorFlow and no released checkpoint is available in this environment.  Pinned without TensorFlow:
crc32c against the published check value, the table / proto decoding against bytes hand-assembled from the format
specifications inside the test (tests/test_cpu_tf_ckpt.py), block and tensor checksums verified on read.  Still
unpinned: that TensorFlow 2.2's writer makes no choice this reader does not expect (e.g. a compressed index)."""
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


_CRC_TABLE = None


def crc32c(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82f63b78): the checksum of table blocks and tensor payloads."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
            tab.append(c)
        _CRC_TABLE = np.array(tab, np.uint32)
    c = 0xffffffff
    tab = _CRC_TABLE
    for b in bytes(data):
        c = int(tab[(c ^ b) & 0xff]) ^ (c >> 8)
    return c ^ 0xffffffff


def masked_crc(data):
    """LevelDB / TensorFlow crc masking: rotate right by 15 bits and add a constant."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def _block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed SSTable block (type %d): the bundle writer emits none" % ctype)
    if verify:
        want = struct.unpack_from('<I', data, offset + size + 1)[0]
        if masked_crc(data[offset:offset + size + 1]) != want:
            raise ValueError("SSTable block at %d: checksum mismatch (corrupt index file)" % offset)
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
            'offset': msg.get(4, [0])[0], 'size': msg.get(5, [0])[0], 'crc': msg.get(6, [None])[0]}


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


def read_string_tensor(prefix, name):
    """Bytes of a scalar DT_STRING entry (e.g. `_CHECKPOINTABLE_OBJECT_GRAPH`) or None: the payload is a varint
    length followed by the bytes (tensor_bundle.cc WriteStringTensor: lengths, a 4-byte length checksum, the bytes)."""
    entries, header = read_index(prefix)
    e = entries.get(name)
    if e is None or e['dtype'] != 7:
        return None
    n_shards = header.get(1, [1])[0] if header else 1
    with open('%s.data-%05d-of-%05d' % (prefix, e['shard_id'], n_shards), 'rb') as h:
        h.seek(e['offset'])
        raw = h.read(e['size'])
    n, pos = _varint(raw, 0)
    if pos + 4 + n == len(raw):      # TF >= 1.x layout: [varint length][4-byte masked crc of the lengths][bytes]
        pos += 4
    return raw[pos:pos + n]


def parse_object_graph(blob):
    """TrackableObjectGraph -> list of nodes: {'children': {local_name: node_id}, 'attributes': {name: checkpoint_key}}."""
    nodes = []
    for raw in _parse_message(blob).get(1, []):
        msg = _parse_message(raw)
        children, attrs = {}, {}
        for c in msg.get(1, []):
            cm = _parse_message(c)
            children[cm.get(2, [b''])[0].decode()] = cm.get(1, [0])[0]
        for a in msg.get(2, []):
            am = _parse_message(a)
            attrs[am.get(1, [b''])[0].decode()] = am.get(3, [b''])[0].decode()
        nodes.append({'children': children, 'attributes': attrs})
    return nodes


def resolve_path(nodes, path):
    """Checkpoint key of the variable reached from the root by the attribute path `a/b/c` (None if there is none)."""
    node = 0
    for part in path.split('/'):
        nxt = nodes[node]['children'].get(part)
        if nxt is None:
            return None
        node = nxt
    return nodes[node]['attributes'].get('VARIABLE_VALUE')


def load_tensors(prefix, names=None, verify=False):
    """{name: np.ndarray} for the numeric tensors of the checkpoint (strings, e.g. the object graph, are skipped).
    verify: check every tensor's crc32c (pure-Python loop: ~1 s per MiB; off by default, on in the tests)."""
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
        if verify and e.get('crc') is not None and masked_crc(raw.tobytes()) != e['crc']:
            raise ValueError("tensor %s: checksum mismatch (corrupt data shard)" % name)
        if e['dtype'] == 14:                           # bfloat16 -> float32
            arr = (raw.view('<u2').astype(np.uint32) << 16).view(np.float32)
        else:
            arr = raw.view(np.dtype(DTYPES[e['dtype']]).newbyteorder('<'))
        out[name] = np.array(arr).reshape(e['shape'])
    return out


def is_tf_checkpoint(path):
    return bool(glob.glob(glob.escape(path) + '.index'))


def to_state_dict(tensors, root='net', graph=None, wanted=None):
    """Checkpoint tensors -> names of a torch state_dict of the mirrored model.
    With the object graph (`graph` = parse_object_graph(...)) and the model's own parameter names (`wanted`, e.g.
    'net_coarse_enc_layer0.kernel'), every name is resolved by walking root/<attr>/<attr>... to its node: this finds
    the tensor whatever key the saver chose for it.  Without a graph: plain key matching,
    'net/net_coarse_enc_layer0/kernel/.ATTRIBUTES/VARIABLE_VALUE' -> 'net_coarse_enc_layer0.kernel'."""
    out = {}
    if graph is not None and wanted is not None:
        for name in wanted:
            key = resolve_path(graph, root + '/' + name.replace('.', '/'))
            if key is not None and key in tensors:
                out[name] = tensors[key]
    pref = root + '/'
    for name, arr in tensors.items():
        if not name.startswith(pref) or not name.endswith(SUFFIX):
            continue
        out.setdefault(name[len(pref):-len(SUFFIX)].replace('/', '.'), arr)
    return out
