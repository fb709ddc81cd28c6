"""TFRecord framing + tf.train.Example wire codec, TF-free.

The reference stores one TFRecord file per block at `tf_record_partial_path.format(block)`
(ecog2txt/data_generators.py:317-326) holding serialized `tf.train.Example`s made by
`tfh.make_feature_example(example_dict)`; float sequences (`ecog_sequence`, `audio_sequence`)
are FLATTENED variable-length float32 lists, all others variable-length byte strings
(subjects.py:297-302; reshape note trainers.py:865).

File framing (TensorFlow's record writer): u64 length | u32 masked-crc32c(length) | payload |
u32 masked-crc32c(payload), little endian, mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8.
Example = message{ Features features = 1 }, Features = map<string, Feature> feature = 1,
Feature = oneof{ BytesList 1, FloatList 2, Int64List 3 }, *List = repeated value = 1."""
import ctypes
import struct

import numpy as np

# ---- crc32c (Castagnoli), table driven --------------------------------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)
_TABLE = np.array(_TABLE, dtype=np.uint32)


def crc32c_python(data):
    """Table-driven reference (one byte per iteration: small inputs and tests only)."""
    crc = 0xFFFFFFFF
    tbl = _TABLE
    for b in bytes(data):
        crc = int(tbl[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_c_crc = None


def crc32c(data):
    """CRC-32C of a bytes-like object or array.  A real subject's records are ~0.4 MB per utterance, so the checksum
    runs in C (e2t_crc32c of the shared library, SSE4.2: GB/s); without the built library, the lane-parallel NumPy
    version of tf_checkpoint (host-side file framing has no bearing on the device path)."""
    global _c_crc
    if _c_crc is None:
        try:
            from . import hip_lib
            _c_crc = hip_lib.load().e2t_crc32c
        except Exception:
            _c_crc = False
    if _c_crc:
        if isinstance(data, np.ndarray):
            buf = np.ascontiguousarray(data)
            return int(_c_crc(buf.ctypes.data, buf.nbytes, 0))
        b = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
        return int(_c_crc((ctypes.c_char * len(b)).from_buffer_copy(b) if isinstance(b, bytearray) else b, len(b), 0))
    from .tf_checkpoint import crc32c_numpy
    return crc32c_numpy(data)


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire helpers ---------------------------------------------------
def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(example_dict):
    """dict name -> (float ndarray | list of bytes | int ndarray)  =>  serialized tf.train.Example."""
    feats = b''
    for name in sorted(example_dict):
        val = example_dict[name]
        if isinstance(val, np.ndarray) and val.dtype.kind == 'f':
            packed = np.ascontiguousarray(val, dtype='<f4').reshape(-1).tobytes()
            feature = _ld(2, _ld(1, packed))                       # FloatList, packed
        elif isinstance(val, np.ndarray) and val.dtype.kind in 'iu':
            packed = b''.join(_varint(int(v) & 0xFFFFFFFFFFFFFFFF) for v in val.reshape(-1))
            feature = _ld(3, _ld(1, packed))
        else:
            items = [v if isinstance(v, bytes) else str(v).encode('utf-8') for v in val]
            feature = _ld(1, b''.join(_ld(1, v) for v in items))   # BytesList
        entry = _ld(1, name.encode('utf-8')) + _ld(2, feature)
        feats += _ld(1, entry)
    return _ld(1, feats)


def _iter_fields(buf, pos, end):
    while pos < end:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 2:
            n, pos = _read_varint(buf, pos)
            yield field, wt, buf[pos:pos + n]
            pos += n
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield field, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported wire type %d' % wt)


def decode_example(payload):
    """serialized tf.train.Example => dict name -> float32 ndarray | list[bytes] | int64 ndarray."""
    payload = memoryview(payload).tobytes()
    out = {}
    for f, _, features in _iter_fields(payload, 0, len(payload)):
        if f != 1:
            continue
        for f2, _, entry in _iter_fields(features, 0, len(features)):
            if f2 != 1:
                continue
            name, feature = None, b''
            for f3, _, v in _iter_fields(entry, 0, len(entry)):
                if f3 == 1:
                    name = v.decode('utf-8')
                elif f3 == 2:
                    feature = v
            value = []
            for kind, _, lst in _iter_fields(feature, 0, len(feature)):
                if kind == 1:
                    value = [v for ff, _, v in _iter_fields(lst, 0, len(lst)) if ff == 1]
                elif kind == 2:
                    chunks = []
                    for ff, wt, v in _iter_fields(lst, 0, len(lst)):
                        if ff == 1:
                            chunks.append(np.frombuffer(v, dtype='<f4'))
                    value = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
                elif kind == 3:
                    vals = []
                    for ff, wt, v in _iter_fields(lst, 0, len(lst)):
                        if ff != 1:
                            continue
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                vals.append(x)
                        else:
                            vals.append(v)
                    value = np.array(vals, dtype=np.uint64).astype(np.int64)
            out[name] = value
    return out


class TFRecordWriter:
    def __init__(self, path):
        self._f = open(path, 'wb')

    def write(self, payload):
        hdr = struct.pack('<Q', len(payload))
        self._f.write(hdr + struct.pack('<I', masked_crc(hdr)) + payload + struct.pack('<I', masked_crc(payload)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def tf_record_iterator(path, check_crc=True):
    """Yield the serialized payloads of one TFRecord file."""
    with open(path, 'rb') as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) < 8:
                raise IOError('truncated record header in %s' % path)
            (n,) = struct.unpack('<Q', hdr)
            (c1,) = struct.unpack('<I', f.read(4))
            payload = f.read(n)
            (c2,) = struct.unpack('<I', f.read(4))
            if check_crc and (c1 != masked_crc(hdr) or c2 != masked_crc(payload)):
                raise IOError('corrupt record in %s' % path)
            yield payload
