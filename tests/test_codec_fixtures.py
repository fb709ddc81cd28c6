"""Rows a3 / f1 / f2 against fixtures that were NOT made by this package's own writers: tf.train.Example bytes from
Google's protobuf runtime, a hand-framed TFRecord file, a hand-assembled TensorFlow V2 checkpoint index
(tests/golden/make_codec_fixtures.py; formats: tensorflow/core/example/*.proto, lib/io/record_writer.h,
leveldb table_format.md, protobuf/tensor_bundle.proto)."""
import os

import numpy as np
import pytest

from ecog2txt_amd import tfrecord, tf_checkpoint

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _expected():
    z = np.load(os.path.join(G, 'codec_example_expected.npz'))
    return dict(ecog_sequence=z['ecog_sequence'], trial_ids=z['trial_ids'], text_sequence=bytes(z['text_sequence']).split(b'\n'))


def test_example_bytes_equal_protobufs_and_decode_back():
    payload = open(os.path.join(G, 'codec_example.bin'), 'rb').read()
    want = _expected()
    got = tfrecord.decode_example(payload)
    assert set(got) == set(want)
    np.testing.assert_array_equal(got['ecog_sequence'], want['ecog_sequence'].reshape(-1))      # stored flattened (trainers.py:865)
    assert got['ecog_sequence'].dtype == np.float32
    assert got['text_sequence'] == want['text_sequence']
    np.testing.assert_array_equal(got['trial_ids'], want['trial_ids'])                          # incl. negative int64 (10-byte varints)
    # this package's encoder produces protobuf's deterministic serialisation byte for byte
    mine = tfrecord.encode_example(dict(ecog_sequence=want['ecog_sequence'], text_sequence=want['text_sequence'], trial_ids=want['trial_ids']))
    assert mine == payload


def test_hand_framed_tfrecord_file_is_read_and_rewritten_identically(tmp_path):
    path = os.path.join(G, 'codec_records.tfrecord')
    payload = open(os.path.join(G, 'codec_example.bin'), 'rb').read()
    recs = list(tfrecord.tf_record_iterator(path, check_crc=True))
    assert recs == [payload, b'', payload[:17]]
    out = tmp_path / 'again.tfrecord'
    with tfrecord.TFRecordWriter(str(out)) as w:
        for r in recs:
            w.write(r)
    assert out.read_bytes() == open(path, 'rb').read()
    # a flipped payload byte must be caught by the checksum
    bad = bytearray(open(path, 'rb').read())
    bad[20] ^= 1
    (tmp_path / 'bad.tfrecord').write_bytes(bytes(bad))
    with pytest.raises(IOError):
        list(tfrecord.tf_record_iterator(str(tmp_path / 'bad.tfrecord')))


def test_hand_assembled_v2_checkpoint_is_read(tmp_path):
    prefix = os.path.join(G, 'codec_ckpt')
    want = {k.replace('__', '/'): v for k, v in np.load(os.path.join(G, 'codec_ckpt_expected.npz')).items()}
    assert sorted(tf_checkpoint.list_variables(prefix, with_dtype=True)) == sorted((k, 1, v.shape) for k, v in want.items())
    got = tf_checkpoint.read_checkpoint(prefix, check_crc=True)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == np.float32
        np.testing.assert_array_equal(got[k], want[k])
    # this package's writer reproduces both files bit for bit (same block layout, restart interval, header proto)
    mine = str(tmp_path / 'mine')
    tf_checkpoint.write_checkpoint(mine, want)
    assert open(mine + '.data-00000-of-00001', 'rb').read() == open(prefix + '.data-00000-of-00001', 'rb').read()
    assert open(mine + '.index', 'rb').read() == open(prefix + '.index', 'rb').read()


def test_fixtures_regenerate_identically():
    """protobuf is installed: the committed bytes are what Google's runtime serialises today."""
    pytest.importorskip('google.protobuf')
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_codec_fixtures', os.path.join(G, 'make_codec_fixtures.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    M = mod.build_messages()
    payload, _ = mod.example_fixture(M)
    assert payload == open(os.path.join(G, 'codec_example.bin'), 'rb').read()
    index, data, _ = mod.checkpoint_fixture(M)
    assert index == open(os.path.join(G, 'codec_ckpt.index'), 'rb').read()
    # the three CRC-32C implementations agree with the fixture's bit-by-bit one
    blob = bytes(range(256)) * 5
    assert mod.crc32c_bitwise(blob) == tfrecord.crc32c(blob) == tfrecord.crc32c_python(blob) == tf_checkpoint.crc32c_numpy(np.frombuffer(blob * 60, np.uint8)[:len(blob)])
