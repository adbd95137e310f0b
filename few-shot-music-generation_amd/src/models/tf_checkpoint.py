"""Reader (and writer) of TensorFlow "tensor bundle" checkpoints -- the files `tf.train.Saver` writes for the reference
(`/root/reference/src/models/tf_model.py:96-97,106-114`: `<dir>/<name>/<name>-<step>.index`, `.data-00000-of-00001`, and the
`checkpoint` state file that `tf.train.latest_checkpoint` reads, `tf_model.py:116-120`).  No TensorFlow needed.

Format (TensorFlow `core/util/tensor_bundle` + its LevelDB-derived `core/lib/io/table*`):
  * `<prefix>.index` is an SSTable: data blocks of prefix-compressed (shared, unshared, value_len, key_delta, value) entries
    followed by a restart array; every block has a 5-byte trailer (compression type: 0 none / 1 snappy, masked crc32c); an index
    block maps "last key of block" -> BlockHandle(offset, size); a 48-byte footer holds the metaindex and index handles and the
    magic 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key is a tensor name whose value is a
    BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6}.
  * `<prefix>.data-<shard>-of-<n>` holds the raw little-endian tensor bytes.

The reference has no checkpoint fixture and TensorFlow is not installable here, so this module is pinned by round trips
through its own writer (plain and snappy-compressed blocks, crc checks; tests/test_tf_checkpoint.py), not by a file TensorFlow
wrote: treat a failure on a real checkpoint as a bug in this reader, not in the file.
"""
import os
import re
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8')}      # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64
DTYPE_IDS = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}


# ------------------------------------------------------------------------------------------------ primitives
def _crc32c_table():
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _crc32c_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in memoryview(data).tobytes() if not isinstance(data, (bytes, bytearray)) else data:
        crc = _TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def snappy_decompress(src):
    """Raw snappy block format (what LevelDB-style tables store): varint uncompressed length, then literal / copy elements."""
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                  # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                  # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:                                # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:                                          # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off <= 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):                            # byte-wise: source and destination may overlap
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch')
    return bytes(out)


# ------------------------------------------------------------------------------------------------ protobuf (the two messages needed)
def _proto_fields(buf):
    """-> list of (field number, wire type, value) of one message (value: int for varint / fixed, bytes for length-delimited)"""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from('<Q', buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            v, pos = struct.unpack_from('<I', buf, pos)[0], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((field, wt, v))
    return out


def _parse_entry(buf):
    entry = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, _, v in _proto_fields(buf):
        if field == 1:
            entry['dtype'] = v
        elif field == 2:
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, x in _proto_fields(dim):
                        if f3 == 1:
                            size = x
                    entry['shape'].append(size)
        elif field == 3:
            entry['shard_id'] = v
        elif field == 4:
            entry['offset'] = v
        elif field == 5:
            entry['size'] = v
        elif field == 6:
            entry['crc32c'] = v
        elif field == 7:
            entry['sliced'] = True
    return entry


def _entry_bytes(dtype_id, shape, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    return (b'\x08' + _put_varint(dtype_id) + b'\x12' + _put_varint(len(dims)) + dims + b'\x20' + _put_varint(offset) +
            b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc))


# ------------------------------------------------------------------------------------------------ table (SSTable)
def _read_block(buf, offset, size):
    body, ctype = buf[offset:offset + size], buf[offset + size]
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if masked_crc(bytes(body) + bytes([ctype])) != stored:
        raise ValueError('block checksum mismatch in checkpoint index')
    if ctype == 1:
        body = snappy_decompress(bytes(body))
    elif ctype != 0:
        raise ValueError('unknown block compression %d' % ctype)
    n_restarts = struct.unpack_from('<I', body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(body, pos)
        unshared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path):
    """-> {tensor name: entry dict} of a `.index` file"""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad magic)' % path)
    footer = buf[len(buf) - 48:]
    _, p = _varint(footer, 0)
    _, p = _varint(footer, p)                          # metaindex handle (unused)
    ioff, p = _varint(footer, p)
    isize, p = _varint(footer, p)
    entries = {}
    for _, handle in _read_block(buf, ioff, isize):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        for key, value in _read_block(buf, boff, bsize):
            if key:                                     # key "" is the BundleHeaderProto
                entries[key.decode()] = _parse_entry(value)
    return entries


VERIFY_LIMIT = 4 << 20        # bytes: the pure-Python crc32c runs at ~10 MB/s, so by default only tensors up to this size are checked


def read_bundle(prefix, verify='auto'):
    """-> {tensor name: ndarray} of the checkpoint `<prefix>.index` + `<prefix>.data-*`.  verify: True checks every tensor's
    crc32c, 'auto' those up to VERIFY_LIMIT bytes (the index blocks are always checked), False none."""
    entries = read_index(prefix + '.index')
    n_shards = max([e['shard_id'] for e in entries.values()] + [0]) + 1
    directory, base = os.path.split(prefix)
    shards = {}
    for name in os.listdir(directory or '.'):
        m = re.match(re.escape(base) + r'\.data-(\d{5})-of-(\d{5})$', name)
        if m:
            shards[int(m.group(1))] = os.path.join(directory, name)
    out = {}
    for name, e in entries.items():
        if e['sliced']:
            raise ValueError('partitioned variable %s: not supported' % name)
        if e['dtype'] not in DTYPES:
            continue                                    # e.g. strings: nothing the model needs
        if e['shard_id'] not in shards:
            raise ValueError('data shard %d of %d missing for %s' % (e['shard_id'], n_shards, prefix))
        with open(shards[e['shard_id']], 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        arr = np.frombuffer(raw, DTYPES[e['dtype']]).reshape(e['shape']).copy()
        check = verify is True or (verify == 'auto' and len(raw) <= VERIFY_LIMIT)
        if check and e['crc32c'] is not None and masked_crc(raw) != e['crc32c']:
            raise ValueError('tensor %s: checksum mismatch' % name)
        out[name] = arr
    return out


def _block(entries, compress=False):
    """entries: sorted [(key bytes, value bytes)] -> block contents + trailer (restart interval 16, like TensorFlow's tables)"""
    body, restarts, prev = bytearray(), [], b''
    for i, (key, value) in enumerate(entries):
        shared = 0
        if i % 16 == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev = key
    for r in restarts or [0]:
        body += struct.pack('<I', r)
    body += struct.pack('<I', max(len(restarts), 1))
    ctype = 0
    if compress:                                        # literal-only snappy stream (valid, if not small): exercises the reader's snappy path
        raw, body, pos = bytes(body), bytearray(_put_varint(len(body))), 0
        while pos < len(raw):
            chunk = raw[pos:pos + 60]
            body += bytes([(len(chunk) - 1) << 2]) + chunk
            pos += len(chunk)
        ctype = 1
    body = bytes(body)
    return body + bytes([ctype]) + struct.pack('<I', masked_crc(body + bytes([ctype])))


def write_bundle(prefix, tensors, block_entries=16, compress=False):
    """Writes {name: ndarray} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard) and registers it in the
    directory's `checkpoint` state file, the way tf.train.Saver.save does."""
    items = sorted(((k.encode(), np.asarray(v, order='C')) for k, v in tensors.items()), key=lambda kv: kv[0])
    data, entries = bytearray(), [(b'', b'\x08\x01\x1a\x02\x08\x01')]        # BundleHeaderProto: num_shards 1, little endian, version {producer 1}
    for key, arr in items:
        if arr.dtype not in DTYPE_IDS:
            raise ValueError('unsupported dtype %s for %s' % (arr.dtype, key))
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        entries.append((key, _entry_bytes(DTYPE_IDS[arr.dtype], arr.shape, len(data), len(raw), masked_crc(raw))))
        data += raw
    out, index = bytearray(), []
    for i in range(0, len(entries), block_entries):
        chunk = entries[i:i + block_entries]
        blk = _block(chunk, compress)
        index.append((chunk[-1][0], _put_varint(len(out)) + _put_varint(len(blk) - 5)))
        out += blk
    meta = _block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta) - 5)
    out += meta
    idx = _block(index)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx) - 5)
    out += idx
    footer = meta_handle + idx_handle
    out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
    with open(os.path.join(os.path.dirname(prefix), 'checkpoint'), 'w') as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by `<directory>/checkpoint`, else the `.index` with the largest step"""
    state = os.path.join(directory, 'checkpoint')
    if os.path.isfile(state):
        with open(state) as f:
            m = re.search(r'^model_checkpoint_path:\s*"(.*)"', f.read(), re.M)
        if m:
            prefix = m.group(1) if os.path.isabs(m.group(1)) else os.path.join(directory, m.group(1))
            if os.path.isfile(prefix + '.index'):
                return prefix
    best = None
    if os.path.isdir(directory):
        for name in os.listdir(directory):
            m = re.match(r'(.*-(\d+))\.index$', name)
            if m and (best is None or int(m.group(2)) > best[0]):
                best = (int(m.group(2)), os.path.join(directory, m.group(1)))
    return best[1] if best else None


# ------------------------------------------------------------------------------------------------ variable names of the reference graph
def map_variables(tensors, n_layers):
    """TF variable names of the reference LSTM baseline (`tf_model.py:91-93`, `lstm_baseline.py:39-40,44-49,60-62,82`) ->
    (params, adam_m, adam_v, global_step) keyed by this repository's names.  Variables live under the model's scope
    `<name>/`; the cell's are `.../multi_rnn_cell/cell_<l>/basic_lstm_cell/{kernel,bias}` (`weights` / `biases` before TF 1.2);
    Adam slots are `<var>/Adam` and `<var>/Adam_1`; the step counter is the scope's unnamed `Variable`."""
    params, m, v, step = {}, {}, {}, None
    for name, arr in tensors.items():
        slot, base = params, name
        if name.endswith('/Adam'):
            slot, base = m, name[:-5]
        elif name.endswith('/Adam_1'):
            slot, base = v, name[:-7]
        leaf = base.rsplit('/', 1)[-1]
        key = None
        if leaf in ('embedding', 'softmax_w', 'softmax_b'):
            key = leaf
        else:
            cell = re.search(r'cell_(\d+)/', base)
            layer = int(cell.group(1)) if cell else (0 if n_layers == 1 else None)
            if layer is not None and leaf in ('kernel', 'weights'):
                key = 'kernel_%d' % layer
            elif layer is not None and leaf in ('bias', 'biases'):
                key = 'bias_%d' % layer
        if key is not None:
            slot[key] = arr
        elif slot is params and leaf in ('Variable', 'global_step') and arr.ndim == 0:
            step = int(arr)
    return params, m, v, step
