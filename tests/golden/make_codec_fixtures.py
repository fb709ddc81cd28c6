"""Independent fixtures for the on-disk formats of rows a3 / f1 / f2.

Nothing here imports ecog2txt_amd: the protocol buffers are encoded by Google's `protobuf` runtime from message
descriptors built at run time after the published .proto definitions (tensorflow/core/example/{example,feature}.proto;
tensorflow/core/protobuf/tensor_bundle.proto, framework/{tensor_shape,types,versions}.proto), the TFRecord frame and the
LevelDB-format table block are assembled by hand from their format documents (tensorflow/core/lib/io/record_writer.h:
u64 length | masked crc32c(length) | payload | masked crc32c(payload); leveldb/doc/table_format.md: prefix-compressed
entries, restart array, 1-byte compression type + masked crc32c trailer, 48-byte footer with the magic number), and the
CRC-32C is a bit-by-bit loop over the Castagnoli polynomial.

    python tests/golden/make_codec_fixtures.py      # rewrites tests/golden/codec_*.bin / .npz
"""
import os
import struct

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
T = descriptor_pb2.FieldDescriptorProto


def crc32c_bitwise(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(data):
    c = crc32c_bitwise(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def build_messages():
    fd = descriptor_pb2.FileDescriptorProto(name='e2t_fixture.proto', package='e2tfix', syntax='proto3')

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, typ, label=T.LABEL_OPTIONAL, type_name=None, oneof=None, packed=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, number, typ, label
        if type_name:
            f.type_name = '.e2tfix.' + type_name
        if oneof is not None:
            f.oneof_index = oneof
        if packed is not None:
            f.options.packed = packed
        return f
    m = msg('BytesList'); field(m, 'value', 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    m = msg('FloatList'); field(m, 'value', 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    m = msg('Int64List'); field(m, 'value', 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    m = msg('Feature')
    m.oneof_decl.add().name = 'kind'
    field(m, 'bytes_list', 1, T.TYPE_MESSAGE, type_name='BytesList', oneof=0)
    field(m, 'float_list', 2, T.TYPE_MESSAGE, type_name='FloatList', oneof=0)
    field(m, 'int64_list', 3, T.TYPE_MESSAGE, type_name='Int64List', oneof=0)
    m = msg('Features')
    e = m.nested_type.add()
    e.name = 'FeatureEntry'
    e.options.map_entry = True
    field(e, 'key', 1, T.TYPE_STRING)
    f = field(e, 'value', 2, T.TYPE_MESSAGE)
    f.type_name = '.e2tfix.Feature'
    f = field(m, 'feature', 1, T.TYPE_MESSAGE, T.LABEL_REPEATED)
    f.type_name = '.e2tfix.Features.FeatureEntry'
    m = msg('Example'); field(m, 'features', 1, T.TYPE_MESSAGE, type_name='Features')
    # tensor bundle (V2 checkpoint index values)
    m = msg('Dim'); field(m, 'size', 1, T.TYPE_INT64); field(m, 'name', 2, T.TYPE_STRING)
    m = msg('TensorShapeProto'); field(m, 'dim', 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name='Dim'); field(m, 'unknown_rank', 3, T.TYPE_BOOL)
    m = msg('VersionDef'); field(m, 'producer', 1, T.TYPE_INT32); field(m, 'min_consumer', 2, T.TYPE_INT32)
    m = msg('BundleHeaderProto'); field(m, 'num_shards', 1, T.TYPE_INT32); field(m, 'endianness', 2, T.TYPE_INT32)
    field(m, 'version', 3, T.TYPE_MESSAGE, type_name='VersionDef')
    m = msg('BundleEntryProto'); field(m, 'dtype', 1, T.TYPE_INT32); field(m, 'shape', 2, T.TYPE_MESSAGE, type_name='TensorShapeProto')
    field(m, 'shard_id', 3, T.TYPE_INT32); field(m, 'offset', 4, T.TYPE_INT64); field(m, 'size', 5, T.TYPE_INT64)
    field(m, 'crc32c', 6, T.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('e2tfix.' + n))
    return {n: get(n) for n in ('Example', 'BundleHeaderProto', 'BundleEntryProto')}


def example_fixture(M):
    rng = np.random.default_rng(7)
    ecog = rng.standard_normal((5, 4)).astype(np.float32)            # [T, C], stored flattened (trainers.py:865)
    words = [b'the_', b'cat_', 'café_'.encode('utf-8')]
    ints = np.array([0, 1, -1, 2 ** 40, -2 ** 62], np.int64)
    ex = M['Example']()
    ex.features.feature['ecog_sequence'].float_list.value.extend(ecog.reshape(-1).tolist())
    ex.features.feature['text_sequence'].bytes_list.value.extend(words)
    ex.features.feature['trial_ids'].int64_list.value.extend(int(v) for v in ints)
    payload = ex.SerializeToString(deterministic=True)               # map entries in key order
    return payload, dict(ecog_sequence=ecog, text_sequence=np.array(words, dtype=object), trial_ids=ints)


def tfrecord_file(payloads):
    out = b''
    for p in payloads:
        hdr = struct.pack('<Q', len(p))
        out += hdr + struct.pack('<I', masked(hdr)) + p + struct.pack('<I', masked(p))
    return out


def table_block(entries, restart_interval=16):
    """leveldb data block: (shared, non_shared, value_len varints | key delta | value)*, restart offsets, count."""
    buf, restarts, last = bytearray(), [0], b''          # (leveldb's BlockBuilder starts with one restart point at 0)
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            if i:
                restarts.append(len(buf))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        buf += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    return bytes(buf) + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))


def checkpoint_fixture(M):
    """A two-variable V2 checkpoint: `<prefix>.index` (one data block) + `<prefix>.data-00000-of-00001`."""
    rng = np.random.default_rng(11)
    arrays = {'seq2seq/decoder_rnn/cell_0/bias': rng.standard_normal(8).astype(np.float32),
              'seq2seq/decoder_rnn/cell_0/kernel': rng.standard_normal((3, 8)).astype(np.float32)}
    data, entries = b'', []
    hdr = M['BundleHeaderProto'](num_shards=1, endianness=0)
    hdr.version.producer = 1
    entries.append((b'', hdr.SerializeToString()))
    for name in sorted(arrays):
        raw = arrays[name].astype('<f4').tobytes()
        e = M['BundleEntryProto'](dtype=1, shard_id=0, offset=len(data), size=len(raw), crc32c=masked(raw))
        for d in arrays[name].shape:
            e.shape.dim.add().size = d
        entries.append((name.encode(), e.SerializeToString()))
        data += raw
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b'\x00' + struct.pack('<I', masked(block + b'\x00')))       # type 0 = uncompressed
        return varint(off) + varint(len(block))
    h_data = emit(table_block(entries))
    h_meta = emit(table_block([]))
    h_index = emit(table_block([(entries[-1][0], h_data)], restart_interval=1))
    foot = h_meta + h_index
    out.extend(foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', 0xdb4775248b80fb57))
    return bytes(out), data, arrays


def main():
    M = build_messages()
    payload, want = example_fixture(M)
    open(os.path.join(HERE, 'codec_example.bin'), 'wb').write(payload)
    open(os.path.join(HERE, 'codec_records.tfrecord'), 'wb').write(tfrecord_file([payload, b'', payload[:17]]))
    np.savez(os.path.join(HERE, 'codec_example_expected.npz'), ecog_sequence=want['ecog_sequence'], trial_ids=want['trial_ids'],
             text_sequence=np.frombuffer(b'\n'.join(want['text_sequence']), np.uint8))
    index, data, arrays = checkpoint_fixture(M)
    open(os.path.join(HERE, 'codec_ckpt.index'), 'wb').write(index)
    open(os.path.join(HERE, 'codec_ckpt.data-00000-of-00001'), 'wb').write(data)
    np.savez(os.path.join(HERE, 'codec_ckpt_expected.npz'), **{k.replace('/', '__'): v for k, v in arrays.items()})
    print('wrote %d-byte example, %d-byte index' % (len(payload), len(index)))


if __name__ == '__main__':
    main()
