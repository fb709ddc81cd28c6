"""TensorFlow V2 checkpoint codec: known answers of the primitives, structure of the
written files, round trips; no TensorFlow needed (and none is available to cross-read)."""
import os
import struct
import time

import numpy as np
import pytest

from ecog2txt_amd import tf_checkpoint as T
from ecog2txt_amd.tfrecord import crc32c_python as crc_serial


def test_crc32c_known_answers_and_lane_combination():
    assert T.crc32c(b'123456789') == 0xE3069283                 # Castagnoli check value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                     # RFC 3720 B.4: 32 bytes of zeros
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43            # RFC 3720 B.4: 32 bytes of ones
    rng = np.random.default_rng(0)
    for n in (4096 * 64, 4096 * 64 + 13, 700001):                # the multi-lane path and its tail
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(b) == T.crc32c_numpy(b) == crc_serial(b)        # C (SSE4.2), NumPy lanes, table-driven Python
    # TensorFlow / LevelDB masking: rotate right by 15, add the delta
    assert T.mask_crc(0) == 0xA282EAD8 and T.mask_crc(0xE3069283) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xA282EAD8) & 0xFFFFFFFF


def test_snappy_literal_and_copies():
    # "abcdabcdabcdXY": literal 'abcd', copy(offset 4, len 8) with a 1-byte offset tag, literal 'XY'
    stream = bytes([14]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([(2 - 1) << 2]) + b'XY'
    assert T._snappy_decompress(stream) == b'abcdabcdabcdXY'
    # 2-byte-offset copy
    stream = bytes([12]) + bytes([(6 - 1) << 2]) + b'abcdef' + bytes([((6 - 1) << 2) | 2, 6, 0])
    assert T._snappy_decompress(stream) == b'abcdefabcdef'


def test_table_round_trip_prefix_compression_and_blocks(tmp_path):
    keys = sorted({('seq2seq/encoder_rnn_%d/%s/cell_0/%s' % (i % 7, d, k)).encode() for i in range(40) for d in ('fw', 'bw')
                   for k in ('kernel', 'bias', 'kernel/Adam', 'kernel/Adam_1')} | {b''})
    items = [(k, bytes([i % 251]) * (i % 37 + 1)) for i, k in enumerate(keys)]
    path = str(tmp_path / 't.index')
    T.write_table(path, items, block_size=256)                   # many small blocks -> a multi-entry index block
    assert T.read_table(path) == items
    buf = open(path, 'rb').read()
    assert struct.unpack('<Q', buf[-8:])[0] == 0xdb4775248b80fb57 and len(buf) > 48
    # a flipped byte in a data block is caught by the block checksum
    bad = bytearray(buf); bad[10] ^= 1
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        T.read_table(path)
    with pytest.raises(ValueError, match='increasing'):
        T.write_table(path, [(b'b', b''), (b'a', b'')])


def test_checkpoint_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(1)
    arrays = {
        'seq2seq/subnet_401/encoder_embedding_256_100_0/weights': rng.standard_normal((1, 12, 256, 100)).astype(np.float32),
        'seq2seq/subnet_401/encoder_embedding_256_100_0/biases': rng.standard_normal(100).astype(np.float32),
        'seq2seq/encoder_rnn_0/fw/cell_0/kernel': rng.standard_normal((500, 1600)).astype(np.float32),
        'seq2seq/encoder_rnn_0/fw/cell_0/kernel/ExponentialMovingAverage': rng.standard_normal((500, 1600)).astype(np.float32),
        'global_step': np.array(1234, np.int64),
        'beta1_power': np.array(0.5, np.float32),
        'empty': np.zeros((0, 3), np.float32),
    }
    prefix = str(tmp_path / 'model.ckpt-7')
    T.write_checkpoint(prefix, arrays)
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(a.nbytes for a in arrays.values())
    got = T.read_checkpoint(prefix, check_crc=True)
    assert set(got) == set(arrays)
    for k, a in arrays.items():
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k
    assert dict(T.list_variables(prefix))['seq2seq/encoder_rnn_0/fw/cell_0/kernel'] == (500, 1600)
    assert set(T.read_checkpoint(prefix, names=['global_step'])) == {'global_step'}
    # header entry: key "" with num_shards = 1; entries carry dtype DT_FLOAT = 1 and the tensor's byte size
    table = dict(T.read_table(prefix + '.index'))
    assert table[b''][:2] == bytes([0x08, 0x01])
    e = T._parse_entry(table[b'seq2seq/encoder_rnn_0/fw/cell_0/kernel'])
    assert e['dtype'] == 1 and e['shape'] == [500, 1600] and e['size'] == 500 * 1600 * 4 and e['shard_id'] == 0
    # corrupting the data file is caught by the per-tensor checksum
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(e['offset'] + 5); f.write(b'\x00\x01\x02')
    with pytest.raises(ValueError, match='checksum'):
        T.read_checkpoint(prefix, check_crc=True)


def test_model_sizes_are_recovered_from_a_tf_checkpoint(tmp_path):
    """recover_model_sizes (trainers.py:444-554) walks names and shapes only: it must work on a TensorFlow checkpoint
    (no .npz next to it), e.g. one written by the reference."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    f32 = lambda *s: np.zeros(s, np.float32)
    arrays = {
        'seq2seq/subnet_401/encoder_embedding_256_100_0/weights': f32(1, 12, 256, 100),
        'seq2seq/subnet_401/encoder_embedding_256_100_0/biases': f32(100),
        'seq2seq/decoder_embedding_1806_150_0/weights': f32(1806, 150),
        'seq2seq/decoder_projection_800_1806_0/weights': f32(1806, 800),
        'seq2seq/encoder_1_projection_800_225_0/weights': f32(800, 225),
        'seq2seq/encoder_1_projection_225_13_1/weights': f32(13, 225),
        'seq2seq/decoder_rnn/cell_0/kernel': f32(950, 3200),
        'seq2seq/decoder_rnn/cell_0/kernel/ExponentialMovingAverage': f32(950, 3200),
    }
    for l, d in enumerate((100, 800, 800)):
        for direction in ('fw', 'bw'):
            arrays['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, direction)] = f32(d + 400, 1600)
    T.write_checkpoint(str(tmp_path / 'model.ckpt-3'), arrays)
    tr = MultiSubjectTrainer.__new__(MultiSubjectTrainer)
    tr._checkpoint_dir, tr._restore_epoch = str(tmp_path), 3
    ls, ds, strides, ema = MultiSubjectTrainer.recover_model_sizes(tr)
    assert ls['encoder_rnn'] == [400, 400, 400] and ls['decoder_rnn'] == [800] and ls['encoder_embedding'] == [100]
    assert ls['decoder_embedding'] == [150] and ls['encoder_1_projection'] == [225] and ls['decoder_projection'] == []
    assert ds['401']['encoder_inputs'] == 256 and ds[None]['decoder_targets'] == 1806 and ds[None]['encoder_1_targets'] == 13
    assert strides['401'] == [12] and ema


def test_a_crashed_writers_temporaries_are_not_taken_for_an_epoch(tmp_path, monkeypatch):
    """The trainer's restore scan keys on 'model.ckpt-<epoch>.index' (trainers.py:240-249): the temporary files of a writer that
    died between writing and renaming must not match it."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    arrays = {'seq2seq/decoder_embedding_10_4_0/weights': np.zeros((10, 4), np.float32)}
    T.write_checkpoint(str(tmp_path / 'model.ckpt-3'), arrays)
    real = os.replace
    monkeypatch.setattr(os, 'replace', lambda a, b: (_ for _ in ()).throw(OSError('killed before the rename')))
    with pytest.raises(OSError):
        T.write_checkpoint(str(tmp_path / 'model.ckpt-9'), arrays)
    monkeypatch.setattr(os, 'replace', real)
    left = [f for f in os.listdir(tmp_path) if '9' in f]
    assert left and all(f.startswith('.tmp-') for f in left)
    tr = MultiSubjectTrainer.__new__(MultiSubjectTrainer)
    tr._checkpoint_dir, tr._restore_epoch = str(tmp_path), None
    assert tr.restore_epoch == 3


def test_backend_checkpoints_store_float32_variables(tmp_path):
    """the reference's TF1 Saver restores into float32 variables and rejects a dtype mismatch, so every
    variable SequenceNetwork._save writes must be DT_FLOAT (1), in the .index as well as in the .npz."""
    import types
    import torch
    from oracle import seq2seq as O
    from helpers import tiny_spec
    from ecog2txt_amd.engine import ParamStore, NetSpec
    from ecog2txt_amd.sequence_network import SequenceNetwork
    ospec = tiny_spec()
    spec = NetSpec(**{k: getattr(ospec, k) for k in NetSpec.__dataclass_fields__})
    store = ParamStore(spec, 'cpu')
    store.import_tf(O.init_params(ospec, seed=2))
    eng = types.SimpleNamespace(store=store, step_t=torch.zeros(1, dtype=torch.int32))
    net = SequenceNetwork.__new__(SequenceNetwork)
    net.checkpoint_path = str(tmp_path / 'model.ckpt')
    net._save(eng, 3)
    prefix = str(tmp_path / 'model.ckpt-3')
    listed = T.list_variables(prefix, with_dtype=True)
    assert listed and all(dtype == 1 for _, dtype, _ in listed), [x for x in listed if x[1] != 1][:3]
    z = np.load(prefix + '.npz')
    assert all(z[k].dtype == np.float32 for k in z.files if not k.startswith('__'))
    back = T.read_checkpoint(prefix)
    k = 'seq2seq/decoder_rnn/cell_0/kernel'
    assert back[k].dtype == np.float32 and np.array_equal(back[k], store.export_tf('p')[k].astype(np.float32))


def test_save_removes_a_dead_writers_temporaries_and_leaves_a_live_writers_alone(tmp_path):
    """the clean-up in front of a checkpoint write must not delete the in-flight temporaries of ANOTHER live process
    writing into the same directory (its os.replace would fail) -- only those whose writer is gone, or this process's own."""
    import subprocess
    import sys
    import types
    import torch
    from oracle import seq2seq as O
    from helpers import tiny_spec
    from ecog2txt_amd.engine import ParamStore, NetSpec
    from ecog2txt_amd.sequence_network import SequenceNetwork
    ospec = tiny_spec()
    spec = NetSpec(**{k: getattr(ospec, k) for k in NetSpec.__dataclass_fields__})
    store = ParamStore(spec, 'cpu')
    store.import_tf(O.init_params(ospec, seed=2))
    eng = types.SimpleNamespace(store=store, step_t=torch.zeros(1, dtype=torch.int32))
    net = SequenceNetwork.__new__(SequenceNetwork)
    net.checkpoint_path = str(tmp_path / 'model.ckpt')
    gone = subprocess.Popen([sys.executable, '-c', 'pass'])
    gone.wait()
    live = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)'])
    try:
        names = {'dead': '.tmp-%d-model.ckpt-7.npz' % gone.pid, 'live': '.tmp-%d-model.ckpt-8.npz' % live.pid,
                 'own': '.tmp-%d-model.ckpt-1.npz' % os.getpid(), 'other': 'notes.txt',
                 # host-tagged names (what the writers use): this host's by process id, another host's by age only -- a live
                 # writer on another node of a shared file system looks dead from here
                 'dead_here': '.tmp-%d-h%s-model.ckpt-7.index' % (gone.pid, T.host_tag()),
                 'elsewhere_fresh': '.tmp-%d-h%s-model.ckpt-9.npz' % (gone.pid, 'f' * 8 if T.host_tag() != 'f' * 8 else '0' * 8),
                 'elsewhere_old': '.tmp-%d-h%s-model.ckpt-6.npz' % (gone.pid, 'f' * 8 if T.host_tag() != 'f' * 8 else '0' * 8)}
        for f in names.values():
            (tmp_path / f).write_bytes(b'x')
        old = time.time() - 7200
        os.utime(tmp_path / names['elsewhere_old'], (old, old))
        net._save(eng, 3)
        left = set(os.listdir(tmp_path))
        assert names['live'] in left and names['other'] in left and names['elsewhere_fresh'] in left
        assert names['dead'] not in left and names['own'] not in left and names['dead_here'] not in left and names['elsewhere_old'] not in left
        assert 'model.ckpt-3.index' in left and 'model.ckpt-3.npz' in left
    finally:
        live.kill()
        live.wait()
