"""Host-side boundary (no GPU): manifest loader, ECoGSubject / SequenceDataManifest / ECoGDataGenerator
surface, TFRecord codec, vocabulary rules, trainer construction.  Expectations are hand-derived from
the cited reference lines (nothing on the reference side can be executed here)."""
import os

import numpy as np
import pytest

import ecog2txt_amd
from ecog2txt_amd import tfrecord
from ecog2txt_amd.manifests import load_manifest
from ecog2txt_amd.subjects import ECoGSubject, SequenceDataManifest
from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
from ecog2txt_amd.toolbox import wer_vector, auto_attribute, str2int_hook
from ecog2txt_amd.sequence_network import target_inds_to_sequences, load_examples
from experiment_fixture import make_experiment

REF = '/root/reference/ecog2txt/auxiliary'


@pytest.fixture
def exp(tmp_path, monkeypatch):
    path = make_experiment(tmp_path, subject_ids=(400, 401))
    monkeypatch.setattr(ECoGDataGenerator, 'text_dir', str(tmp_path))
    return path, tmp_path


def test_constants_match_reference_contract():
    # ecog2txt/__init__.py:13-22
    assert (ecog2txt_amd.EOS_token, ecog2txt_amd.pad_token, ecog2txt_amd.OOV_token) == ('<EOS>', '<pad>', '<OOV>')
    assert ecog2txt_amd.DATA_PARTITIONS == {'training', 'validation', 'testing'}
    assert 'word_sequence' in ecog2txt_amd.TOKEN_TYPES and len(ecog2txt_amd.TOKEN_TYPES) == 6


def test_manifest_python_tags(exp):
    m = load_manifest(exp[0])
    assert set(m) == {400, 401}
    assert m[401]['DataGenerator'] is SyntheticSpeechDataGenerator
    assert m[401]['RGB_color'] == (0.4, 0.65, 0.11)
    assert m[401]['block_types']['training'] == {'mocha-1', 'mocha-2'}


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_reference_manifests_load_unchanged():
    """The reference's own manifests (which yaml.full_load cannot resolve without TF) load as they are."""
    for name, sids in (('mocha-1_word_sequence.yaml', {400, 401, 402, 403}), ('demo2_word_sequence.yaml', None)):
        m = load_manifest(os.path.join(REF, 'EFC', name))
        if sids:
            assert set(m) == sids
        first = m[sorted(m)[0]]
        assert first['token_type'] == 'word_sequence' and first['layer_sizes']['encoder_rnn'] == [400, 400, 400]
    m = load_manifest(os.path.join(REF, 'EFC', 'mocha-1_word_sequence.yaml'))
    assert m[400]['layer_sizes']['decoder_rnn'] == [800] and m[400]['sampling_rate_decimated'] == 16.5
    assert m[402]['grid_size'] == [8, 16]


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_reference_block_partitions_subject_401():
    """subject 401 with the mocha-1 manifest -> 8 training, 1 validation, 1 testing block."""
    m = load_manifest(os.path.join(REF, 'EFC', 'mocha-1_word_sequence.yaml'))
    man = dict(m[401], json_dir=os.path.join(REF, 'EFC'), DataGenerator=ECoGDataGenerator, bad_electrodes_path=None,
               good_electrodes=None)
    s = ECoGSubject(man, 401)
    ids = s.block_ids
    assert (len(ids['training']), len(ids['validation']), len(ids['testing'])) == (8, 1, 1)
    assert s.decimation_factor == 12                                   # round(200 / 16.5)
    assert s.data_generator.num_ECoG_channels == 480                   # 16x16 bipolar: 16*15 + 15*16
    assert s.data_generator.max_samples == 1250                        # floor(200 * 6.25)
    man402 = dict(m[402], json_dir=os.path.join(REF, 'EFC'), DataGenerator=ECoGDataGenerator, bad_electrodes_path=None)
    s2 = ECoGSubject(man402, 402)
    assert s2.decimation_factor == 12                                  # round(190.73486328125 / 16.5)
    assert s2.data_generator.num_ECoG_channels == 232                  # 8x16 bipolar: 8*15 + 7*16
    # vocab file of the reference: specials first, words carry the trailing underscore
    with open(os.path.join(REF, 'vocab.mocha-timit.1806')) as f:
        v = f.read().split()
    assert len(v) == 1806 and v[:3] == ['<pad>', '<EOS>', '<OOV>'] and all(w.endswith('_') for w in v[3:])


def test_subject_surface(exp):
    m = load_manifest(exp[0])
    s = ECoGSubject(m[401], 401)
    assert s.subnet_id == 401
    ids = s.block_ids
    assert ids == {'training': {1, 2, 3}, 'validation': {4}, 'testing': {5}}
    s_all = ECoGSubject(m[400], 400, pretrain_all_blocks=True)
    assert s_all.block_ids['training'] == {1, 2, 3, 4, 5}
    assert s.decimation_factor == 12
    s.decimation_factor = 7
    assert s.decimation_factor == 7
    assert s.tf_record_partial_path.endswith('SYN401_B{0}.tfrecord')
    dms = s.data_manifests
    assert dms['encoder_inputs'].num_features == 16 and dms['encoder_inputs'].distribution == 'Rayleigh'
    assert dms['encoder_1_targets'].num_features == 5 and dms['encoder_1_targets'].distribution == 'Gaussian'
    assert dms['decoder_targets'].distribution == 'categorical'
    assert dms['encoder_inputs'].padding_value == 0.0


def test_generator_geometry(exp):
    m = load_manifest(exp[0])
    man = dict(m[401], grid_size=[16, 16])
    g = ECoGDataGenerator(man, 401)
    lay = g.elec_layout
    assert np.array_equal(lay, np.arange(255, -1, -1).reshape(16, 16).T)          # data_generators.py:103-109
    assert g.num_ECoG_channels == 256
    assert ECoGDataGenerator(man, 401, REFERENCE_BIPOLAR=True).num_ECoG_channels == 480
    assert ECoGDataGenerator(dict(man, grid_size=[8, 16]), 401, REFERENCE_BIPOLAR=True).num_ECoG_channels == 232
    assert ECoGDataGenerator(man, 401, USE_FIELD_POTENTIALS=True).num_ECoG_channels == 512
    assert g.max_samples == 1250 and ECoGDataGenerator(man, 401, max_samples=300).max_samples == 300
    assert ECoGDataGenerator(man, 401, token_type='word').max_samples == 200
    assert g._sentence_tokenize(['The', 'cat']) == [b'the_', b'cat_']              # data_generators.py:468-473
    # 1-indexed bad electrodes -> 0-indexed good set (data_generators.py:173-193)
    with open(g.bad_electrodes_path, 'w') as f:
        f.write('1\n256\n')
    assert ECoGDataGenerator(man, 401).good_electrodes == set(range(1, 255))
    assert ECoGDataGenerator(dict(man, grid_step=2), 401, REFERENCE_BIPOLAR=True).tf_record_partial_path.split(os.sep)[-2] == 'lowdensity_bipolar'


def test_data_manifest_transforms():
    feats = ['<pad>', '<EOS>', '<OOV>', 'the_', 'cat_']
    dm = SequenceDataManifest('text_sequence', get_feature_list=lambda: feats, APPEND_EOS=True)
    assert dm.num_features == 5 and dm.num_features_raw == 1 and dm.padding_value == 0
    np.testing.assert_array_equal(dm.transform([b'the_', b'dog_', b'cat_']), [3, 2, 4, 1])
    dm.APPEND_EOS = False
    np.testing.assert_array_equal(dm.transform([b'cat_']), [4])
    assert SequenceDataManifest('phoneme_sequence', get_feature_list=lambda: ['a', 'b']).transform([b'zz']).tolist() == [2]


def test_text_helpers():
    feats = ['<pad>', '<EOS>', '<OOV>', 'the_', 'cat_']
    assert target_inds_to_sequences([[3, 4, 1, 0, 0], [4, 1, 0, 0, 0]], feats) == ['the cat', 'cat']
    np.testing.assert_allclose(wer_vector(['the cat sat', 'a'], ['the cat', 'a b']), [1 / 3, 1.0])
    assert str2int_hook({'4': 1, 'x': 2}) == {4: 1, 'x': 2}

    class K:
        @auto_attribute(CHECK_MANIFEST=True)
        def __init__(self, manifest, a=None, b=3, _hidden=5):
            pass
    k = K({'a': 7, 'b': 9})
    assert (k.a, k.b) == (7, 3) and not hasattr(k, '_hidden')


def test_records_roundtrip_and_examples(exp):
    m = load_manifest(exp[0])
    s = ECoGSubject(m[401], 401)
    words = s.write_tf_records_maybe()
    assert all(w.endswith('_') for w in words) and len(words) > 3
    assert os.path.exists(s.tf_record_partial_path.format(5))
    feats = s.data_generator.get_class_list('text_sequence')
    dm = s.data_manifests['decoder_targets']
    dm.get_feature_list = lambda: feats
    dm.APPEND_EOS = True
    ex = load_examples(s, s.block_ids['validation'])
    assert len(ex) == SyntheticSpeechDataGenerator.trials_per_block
    e = ex[0]
    assert e['encoder_inputs'].shape[1] == 16 and e['encoder_1_targets'].shape[1] == 5
    assert e['encoder_inputs'].shape[0] == e['encoder_1_targets'].shape[0]
    assert e['decoder_targets'][-1] == 1 and (e['decoder_targets'][:-1] >= 3).all()
    # same block regenerates identically (deterministic synthetic participant)
    ex2 = list(s.data_generator._ecog_token_generator(4))
    np.testing.assert_array_equal(ex2[0]['ecog_sequence'], e['encoder_inputs'])


def test_trainer_construction_and_vocab_priority(exp):
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    ckpt = str(exp[1] / 'ckpt')
    os.makedirs(ckpt)
    tr = MultiSubjectTrainer(exp[0], [400, 401], checkpoint_dir=ckpt, VERBOSE=False, SN_kwargs={'FF_dropout': 0.4})
    assert tr.net.FF_dropout == 0.4 and tr.net.RNN_dropout == 0.3 and tr.net.N_epochs == 2       # kwarg beats manifest
    assert tr.net.checkpoint_path == os.path.join(ckpt, 'model.ckpt')
    assert tr.ecog_subjects[0].pretrain_all_blocks and not tr.ecog_subjects[1].pretrain_all_blocks
    dm = tr.ecog_subjects[-1].data_manifests['decoder_targets']
    assert dm.APPEND_EOS and dm.get_feature_list()[:3] == ['<pad>', '<EOS>', '<OOV>'] and dm.num_features == 23
    assert tr.ecog_subjects[-1].data_manifests['encoder_1_targets'].penalty_scale == 0.5
    assert tr.restore_epoch is None
    open(os.path.join(ckpt, 'model.ckpt-30.index'), 'w').close()
    open(os.path.join(ckpt, 'model.ckpt-120.index'), 'w').close()
    assert tr.restore_epoch == 120                                                            # trainers.py:235-252
    tr2 = MultiSubjectTrainer(exp[0], [401], checkpoint_dir=ckpt, VERBOSE=False, text_sequence_vocab_list=['<pad>', '<EOS>', '<OOV>', 'x_'])
    assert tr2.ecog_subjects[0].data_manifests['decoder_targets'].num_features == 4
    spec = tr.net._spec_from(tr.ecog_subjects)
    assert spec.channels == {400: 16, 401: 16} and spec.decimation == 12 and spec.aux_layer == 1 and spec.aux_dim == 5
    assert spec.aux_scale == 0.5 and spec.enc_rnn == [32, 32] and spec.vocab == 23


def test_vocab_from_records_when_no_file(exp, monkeypatch):
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    monkeypatch.setattr(ECoGDataGenerator, 'text_dir', str(exp[1] / 'nowhere'))
    tr = MultiSubjectTrainer(exp[0], [401], checkpoint_dir=str(exp[1]), VERBOSE=False)
    feats = tr.ecog_subjects[0].data_manifests['decoder_targets'].get_feature_list()
    assert feats[:3] == ['<pad>', '<EOS>', '<OOV>'] and len(feats) > 3 and len(set(feats)) == len(feats)
