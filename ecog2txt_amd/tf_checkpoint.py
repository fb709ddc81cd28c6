"""TensorFlow checkpoint (V2 "tensor bundle") reader and writer without TensorFlow.

The reference saves and restores its networks with TF1's Saver (`model.ckpt-<epoch>.index`
+ `model.ckpt-<epoch>.data-00000-of-00001`, discovered by `MultiSubjectTrainer.restore_epoch`,
ecog2txt/trainers.py:235-252, and walked variable by variable in `recover_model_sizes`, trainers.py:444-554).
This module reads such a pair into `{variable name: ndarray}` and writes one, so that checkpoints can move between
the reference and this backend in both directions.

Format, restated from the published TensorFlow / LevelDB sources (tensorflow/core/util/tensor_bundle,
tensorflow/core/lib/io/table*, leveldb/doc/table_format.md):

* `<prefix>.index` is an SSTable: data blocks of prefix-compressed entries
  `[varint shared][varint non_shared][varint value_len][key suffix][value]` followed by a restart array
  (`uint32` offsets, then their count); every block has a 5-byte trailer (compression type: 0 none / 1 snappy, then
  the masked CRC-32C of block + type); an (empty) metaindex block; an index block whose values are block handles
  (`varint offset, varint size`); a 48-byte footer = the two handles, zero padding, magic `0xdb4775248b80fb57`.
* Key `""` holds a `BundleHeaderProto` (num_shards, endianness, version), every other key is a variable name whose
  value is a `BundleEntryProto`: dtype (1), shape (2), shard_id (3), offset (4), size (5), masked crc32c (6).
* `<prefix>.data-<shard>-of-<n>` is the raw little-endian row-major bytes of the tensors.

Validation in this repository: CRC-32C and masking against their known-answer vectors, round trips, structural checks.
There is no TensorFlow in the build environment, so no file written by TensorFlow itself was available to read.
"""
import os
import struct

import numpy as np

from .tfrecord import _varint, _read_varint, _iter_fields, _ld, _TABLE

MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xA282EAD8
# tensorflow/core/framework/types.proto
_DTYPES = {1: '<f4', 2: '<f8', 3: '<i4', 4: 'u1', 5: '<i2', 6: 'i1', 9: '<i8', 10: '?', 14: '<u2', 17: '<u2', 19: '<f2',
           22: '<u4', 23: '<u8'}
_DT_OF = {'float32': 1, 'float64': 2, 'int32': 3, 'uint8': 4, 'int16': 5, 'int8': 6, 'int64': 9, 'bool': 10, 'float16': 19,
          'uint32': 22, 'uint64': 23}


# ---- CRC-32C over large buffers: many lanes in numpy, then combined ------------------------------------------------
def _gf2_times(mat, vec):
    s = 0
    i = 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def _crc_shift_op(nbytes):
    """The GF(2) operators whose product advances a (finalised) CRC-32C over `nbytes` following bytes: zlib's
    crc32_combine scheme with the Castagnoli polynomial.  combine(crcA, crcB, len(B)) = apply(ops, crcA) ^ crcB."""
    odd = [0x82F63B78] + [1 << n for n in range(31)]        # operator for one zero bit
    even = _gf2_square(odd)                                  # two zero bits
    odd = _gf2_square(even)                                  # four zero bits
    ops = []
    n = nbytes
    while True:
        even = _gf2_square(odd)                              # first time round: eight zero bits = one byte
        if n & 1:
            ops.append(even)
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            ops.append(odd)
        n >>= 1
        if not n:
            break
    return ops


def crc32c(data):
    """CRC-32C (Castagnoli) of a bytes-like object: the shared library's e2t_crc32c (SSE4.2 instruction) when it is
    built, else crc32c_numpy."""
    from . import tfrecord
    return tfrecord.crc32c(data)


def crc32c_numpy(data):
    """CRC-32C in NumPy: 4096 lanes for large inputs, combined as zlib does (kept as the independent second
    implementation the C one is tested against)."""
    if isinstance(data, np.ndarray):
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = buf.size
    lanes = 4096
    tbl = _TABLE

    def serial(b, crc=0xFFFFFFFF):
        for x in b.tobytes():
            crc = int(tbl[(crc ^ x) & 0xFF]) ^ (crc >> 8)
        return crc
    if n < lanes * 64:
        return serial(buf) ^ 0xFFFFFFFF
    seg = n // lanes
    body = buf[:seg * lanes].reshape(lanes, seg)
    crc = np.full(lanes, 0xFFFFFFFF, np.uint32)
    for j in range(seg):
        crc = tbl[(crc ^ body[:, j]) & 0xFF] ^ (crc >> np.uint32(8))
    crc ^= np.uint32(0xFFFFFFFF)                             # per-lane finalised CRCs
    ops = _crc_shift_op(seg)
    total = int(crc[0])
    for k in range(1, lanes):
        for m in ops:
            total = _gf2_times(m, total)
        total ^= int(crc[k])
    tail = buf[seg * lanes:]
    if tail.size:
        total = serial(tail, total ^ 0xFFFFFFFF) ^ 0xFFFFFFFF
    return total


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- snappy (blocks of a TF-written index may be compressed) -------------------------------------------------------
def _snappy_decompress(src):
    n, pos = _read_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: decompressed %d bytes, header says %d' % (len(out), n))
    return bytes(out)


# ---- SSTable --------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, check_crc=True):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    if check_crc:
        want = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if mask_crc(crc32c(bytes(raw) + bytes([ctype]))) != want:
            raise ValueError('table block at %d: checksum mismatch' % offset)
    if ctype == 1:
        raw = _snappy_decompress(bytes(raw))
    elif ctype != 0:
        raise ValueError('table block at %d: unknown compression type %d' % (offset, ctype))
    return bytes(raw)


def _block_entries(block):
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, check_crc=True):
    """All (key, value) pairs of an SSTable file, in key order."""
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad magic number)' % path)
    foot = buf[-48:]
    pos = 0
    _, pos = _read_varint(foot, pos)
    _, pos = _read_varint(foot, pos)                       # metaindex handle (unused)
    ioff, pos = _read_varint(foot, pos)
    isz, pos = _read_varint(foot, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isz, check_crc)):
        off, p = _read_varint(handle, 0)
        sz, p = _read_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, off, sz, check_crc)))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.ri, self.buf, self.restarts, self.n, self.last = restart_interval, bytearray(), [0], 0, b''

    def add(self, key, value):
        shared = 0
        if self.n % self.ri == 0:
            if self.n:
                self.restarts.append(len(self.buf))
        else:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: iterable of (key bytes, value bytes), keys strictly increasing.  Uncompressed blocks."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return _varint(off) + _varint(len(block))
    index = _BlockBuilder(1)
    cur, last = _BlockBuilder(16), None
    for key, value in items:
        if last is not None and key <= last:
            raise ValueError('table keys must be strictly increasing')
        cur.add(key, value)
        last = key
        if len(cur.buf) >= block_size:
            index.add(last, emit(cur.finish()))
            cur = _BlockBuilder(16)
    if cur.n:
        index.add(last, emit(cur.finish()))
    meta = emit(_BlockBuilder(16).finish())
    idx = emit(index.finish())
    foot = meta + idx
    out.extend(foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', MAGIC))
    with open(path, 'wb') as f:
        f.write(bytes(out))


# ---- bundle protos ----------------------------------------------------------------------------------------------------
def _parse_entry(value):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, wt, v in _iter_fields(value, 0, len(value)):
        if field == 1:
            e['dtype'] = v
        elif field == 2:
            for f2, _, dim in _iter_fields(v, 0, len(v)):
                if f2 == 2:
                    size = 0
                    for f3, _, x in _iter_fields(dim, 0, len(dim)):
                        if f3 == 1:
                            size = x
                    e['shape'].append(size)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = v
        elif field == 5:
            e['size'] = v
        elif field == 6:
            e['crc32c'] = struct.unpack('<I', bytes(v))[0]
        elif field == 7:
            e['sliced'] = True
    return e


def _shard_name(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def read_checkpoint(prefix, check_crc=False, names=None):
    """{variable name: ndarray} of the checkpoint `<prefix>.index` / `<prefix>.data-*` (names: optional filter).
    check_crc also verifies every tensor's CRC-32C (pure Python/NumPy: seconds for tens of MB)."""
    entries = read_table(prefix + '.index')
    num_shards = 1
    out = {}
    shards = {}
    for key, value in entries:
        if key == b'':
            for field, _, v in _iter_fields(value, 0, len(value)):
                if field == 1:
                    num_shards = v
                elif field == 2 and v != 0:
                    raise ValueError('big-endian checkpoints are not supported')
            continue
        name = key.decode('utf-8')
        if names is not None and name not in names:
            continue
        e = _parse_entry(value)
        if e['sliced']:
            raise ValueError('%s: partitioned (sliced) variables are not supported' % name)
        if e['dtype'] not in _DTYPES:
            continue                                        # strings / resources: not weights
        if e['shard_id'] not in shards:
            shards[e['shard_id']] = np.memmap(_shard_name(prefix, e['shard_id'], num_shards), dtype=np.uint8, mode='r')
        raw = shards[e['shard_id']][e['offset']:e['offset'] + e['size']]
        if check_crc and e['crc32c'] is not None and mask_crc(crc32c(np.asarray(raw))) != e['crc32c']:
            raise ValueError('%s: tensor checksum mismatch' % name)
        out[name] = np.frombuffer(bytes(raw), dtype=_DTYPES[e['dtype']]).reshape(e['shape']).copy()
    return out


def list_variables(prefix, with_dtype=False):
    """[(name, shape)] without touching the data files (what recover_model_sizes needs); with_dtype: [(name, TF DataType
    enum (1 = DT_FLOAT, 2 = DT_DOUBLE, 3 = DT_INT32, ...), shape)]."""
    out = []
    for key, value in read_table(prefix + '.index'):
        if key:
            e = _parse_entry(value)
            out.append((key.decode('utf-8'), e['dtype'], tuple(e['shape'])) if with_dtype else (key.decode('utf-8'), tuple(e['shape'])))
    return out


def host_tag():
    """8 hex digits naming this host (CRC-32 of its node name): a process id means something on ITS host only."""
    import platform
    import zlib
    return '%08x' % (zlib.crc32(platform.node().encode('utf-8')) & 0xFFFFFFFF)


def temp_prefix():
    """Name prefix of a checkpoint writer's temporaries: '.tmp-<pid>-h<host tag>-' (SequenceNetwork._save reaps what crashed
    writers left: by process id on the same host, by age for names that carry another host's tag -- a shared file system)."""
    return '.tmp-%d-h%s-' % (os.getpid(), host_tag())


def write_checkpoint(prefix, arrays):
    """Write {name: ndarray} as a single-shard TF V2 checkpoint (`<prefix>.index`, `<prefix>.data-00000-of-00001`)."""
    items = []
    offset = 0
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    # both files are written under temporary names and moved into place, the data shard first and the `.index` -- the file a
    # restore scans for (trainers.py:235-252) -- last: a reader that finds the index finds a complete checkpoint
    # (temporary names start with '.tmp-': the trainer's restore scan keys on names that START with 'model.ckpt-' and end in
    #  '.index' -- a temporary index left behind by a crash must not look like a finished epoch)
    final_prefix = prefix
    prefix = os.path.join(os.path.dirname(prefix), temp_prefix() + os.path.basename(prefix))
    with open(_shard_name(prefix, 0, 1), 'wb') as f:
        for name in sorted(arrays, key=lambda s: s.encode('utf-8')):
            a = np.asarray(arrays[name], order='C')              # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype.name not in _DT_OF:
                raise ValueError('%s: dtype %s cannot be stored' % (name, a.dtype))
            a = a.astype(a.dtype.newbyteorder('<'), copy=False)
            raw = a.tobytes()
            shape = b''.join(_ld(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in a.shape)
            entry = _varint((1 << 3) | 0) + _varint(_DT_OF[a.dtype.name]) + _ld(2, shape)
            if offset:
                entry += _varint((4 << 3) | 0) + _varint(offset)
            entry += _varint((5 << 3) | 0) + _varint(len(raw))
            entry += _varint((6 << 3) | 5) + struct.pack('<I', mask_crc(crc32c(raw)))
            items.append((name.encode('utf-8'), entry))
            f.write(raw)
            offset += len(raw)
    header = _varint((1 << 3) | 0) + _varint(1) + _ld(3, _varint((1 << 3) | 0) + _varint(1))     # num_shards 1, version.producer 1
    write_table(prefix + '.index', [(b'', header)] + items)
    os.replace(_shard_name(prefix, 0, 1), _shard_name(final_prefix, 0, 1))
    os.replace(prefix + '.index', final_prefix + '.index')
