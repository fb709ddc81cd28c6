"""End-to-end on the GPU through the reference-shaped boundary: README flow, transfer-learning
schedules with regex scopes, checkpoint discovery, assessment, saliency."""
import os

import numpy as np
import pytest
import torch

from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
from experiment_fixture import make_experiment

pytestmark = pytest.mark.gpu


@pytest.fixture
def small(tmp_path, monkeypatch):
    monkeypatch.setattr(ECoGDataGenerator, 'text_dir', str(tmp_path))
    monkeypatch.setattr(SyntheticSpeechDataGenerator, 'num_sentences', 6)
    monkeypatch.setattr(SyntheticSpeechDataGenerator, 'trials_per_block', 24)
    monkeypatch.setattr(SyntheticSpeechDataGenerator, 'max_words', 5)
    return tmp_path


def test_readme_flow_learns(small):
    """README.md:72-102 usage, unchanged: construct, write records, parallel_transfer_learn.  Six sentences
    with sentence-dependent ECoG must be learnt almost perfectly."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    path = make_experiment(small, subject_ids=(401,), epochs=60, interval=20)
    ck = str(small / 'ck'); os.makedirs(ck)
    tr = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False,
                             SN_kwargs={'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.1, 'EMA_decay': 0.9},
                             DG_kwargs={'max_samples': 420})
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    a = tr.parallel_transfer_learn()
    wer = a['validation'].decoder_word_error_rates
    acc = a['validation'].decoder_accuracies
    assert len(wer) == 3 and wer[0] > 0.8
    assert a['training'].losses[-1]['decoder'] < 0.5 * a['training'].losses[0]['decoder']
    assert tr.restore_epoch == 60 and os.path.exists(os.path.join(ck, 'model.ckpt-60.index'))
    assert os.path.exists(os.path.join(str(small), 'saved_results'))
    # restore + assess reproduces a finished model; WER must be low
    res = tr.assess_saved_model()
    assert res['validation'].word_error_rate < 0.25, res['validation'].word_error_rate
    assert res['validation'].accuracy > 0.8
    assert res['validation'].decoder_confusions.shape == (23, 23)
    # the same checkpoint assessed with beam search (`beam_width: 4`, `temperature`: manifest keys, mocha-1_word_sequence.yaml:31,
    # 82 -- a fresh trainer with the keys overridden): the trained model is confident, so the best beam is the greedy sequence
    tr_b = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False, restore_epoch=60,
                               SN_kwargs={'N_cases': 32, 'beam_width': 4, 'temperature': 0.8, 'EMA_decay': 0.9}, DG_kwargs={'max_samples': 420})
    res_b = tr_b.assess_saved_model()
    assert tr_b.net.beam_width == 4
    assert res_b['validation'].word_error_rate <= res['validation'].word_error_rate + 0.05
    assert (np.asarray(res_b['validation'].hypotheses, dtype=object) == np.asarray(res['validation'].hypotheses, dtype=object)).mean() > 0.8
    # sizes recovered from the checkpoint by the reference's variable grammar
    ls, ds, strides, ema = tr.recover_model_sizes()
    assert ls['encoder_rnn'] == [32, 32] and ls['decoder_rnn'] == [64] and ls['encoder_embedding'] == [24]
    assert ds['401']['encoder_inputs'] == 16 and ds[None]['decoder_targets'] == 23 and strides['401'] == [12] and ema
    # saliency: gradient w.r.t. the inputs, per electrode and per sample
    sal = tr.get_saliencies('decoder_saliency_map')
    assert sal.shape == (16,) and np.isfinite(sal).all() and sal.max() > 0
    seqs = tr.get_saliencies('decoder_saliency_map', assessment_type='sequences')
    assert seqs.ndim == 3 and seqs.shape[2] == 16
    w = tr.net.get_weights_as_numpy_array('seq2seq/subnet_401/encoder_embedding_16_24_0/weights/ExponentialMovingAverage', 60)
    assert w.shape == (1, 12, 16, 24)
    # activation probe (trainers.py:757-765): shapes, reversal, and the front-end against the oracle's conv on the EMA weights
    act = tr.get_internal_activations()
    n = len(res['validation'].references)
    X = act['reversed_inputs']
    assert X.shape[0] == n and X.shape[2] == 16
    S = X.shape[1] // 12
    assert act['convolved_inputs'].shape == (n, S, 24) and act['final_RNN_state'].shape == (2, n, 64)
    assert act['decimated_reversed_targets'] is None or act['decimated_reversed_targets'].shape[:2] == (n, S)
    from oracle.bf16 import round_bf16
    wc = round_bf16(w.reshape(12 * 16, 24))
    bc = tr.net.get_weights_as_numpy_array('seq2seq/subnet_401/encoder_embedding_16_24_0/biases/ExponentialMovingAverage', 60)
    lens = (np.abs(X).sum(-1) > 0).sum(-1)
    for i in range(min(n, 4)):
        z = round_bf16(X[i]).reshape(S, 12 * 16) @ wc + bc
        z = np.maximum(z, 0.0) * (np.arange(S)[:, None] < -(-lens[i] // 12))
        np.testing.assert_allclose(act['convolved_inputs'][i], round_bf16(z), rtol=2e-2, atol=2e-2)
    assert np.isfinite(act['final_RNN_state']).all() and np.abs(act['final_RNN_state'][1]).max() <= 1.0
    # ... and every probed tensor against the oracle run on the checkpoint's EMA weights (dropout off): front-end output,
    # decimated reversed auxiliary targets, and the (c, h) that initialise the decoder
    from oracle import seq2seq as O
    vals = dict(np.load(os.path.join(ck, 'model.ckpt-60.npz')))
    sfx = '/ExponentialMovingAverage'
    Pema = {k[:-len(sfx)]: v.astype(np.float64) for k, v in vals.items() if k.endswith(sfx)}
    ospec = O.NetSpec(**tr.net._engine.spec.as_dict())
    assert set(Pema) == set(O.init_params(ospec, seed=0))
    vdata = tr.net._stage(tr.ecog_subjects[-1], 'validation')
    obatch = dict(subnet_id=tr.ecog_subjects[-1].subnet_id, encoder_inputs=vdata['X'], decoder_targets=vdata['Y'], encoder_targets=vdata['A'])
    _, oc = O.forward(Pema, ospec, obatch, train=False, emulate_bf16=True)
    np.testing.assert_allclose(act['convolved_inputs'], oc['E'].transpose(1, 0, 2), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(act['final_RNN_state'][0], oc['c0'], rtol=2e-2, atol=5e-3)
    np.testing.assert_allclose(act['final_RNN_state'][1], oc['h0'], rtol=2e-2, atol=5e-3)
    np.testing.assert_allclose(act['decimated_reversed_targets'], oc['aux_heads'][ospec.aux_layer]['At'].transpose(1, 0, 2), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(act['reversed_inputs'], O.reverse_time_major(vdata['X'].astype(np.float64), oc['lens']).transpose(1, 0, 2).astype(np.float32))
    # the assessment itself against the oracle: greedy hypotheses (as text), teacher-forced token accuracy, WER
    from ecog2txt_amd.sequence_network import target_inds_to_sequences
    from ecog2txt_amd.toolbox import wer_vector
    feats = list(tr.ecog_subjects[-1].data_manifests['decoder_targets'].get_feature_list())
    ohyp, _ = O.greedy_decode(Pema, ospec, obatch, max_len=vdata['L'], emulate_bf16=True)
    otext = target_inds_to_sequences(ohyp, feats)
    assert otext == res['validation'].hypotheses
    olo, _ = O.forward(Pema, ospec, obatch, train=False, emulate_bf16=True)
    assert abs(olo['accuracy'] - res['validation'].accuracy) < 1e-6
    assert abs(float(np.mean(wer_vector(res['validation'].references, otext))) - res['validation'].word_error_rate) < 1e-9
    # saliencies (trainers.py:703-732) against the oracle's input gradient of the same penalty-weighted loss
    if vdata['n'] <= 32:                                     # one batch: the loss normalisation is the oracle's
        import dataclasses
        osal = dataclasses.replace(ospec, aux_scale=0.0, dec_scale=1.0)      # get_saliencies: every penalty zeroed but the named one
        _, ocs = O.forward(Pema, osal, obatch, train=False, emulate_bf16=True)
        O.backward(Pema, ocs)
        want = O.input_gradient(Pema, ocs)
        assert seqs.shape == want.shape
        assert np.linalg.norm(seqs - want) / np.linalg.norm(want) < 3e-2
        np.testing.assert_allclose(sal, np.sqrt((want ** 2).mean(axis=(0, 1))), rtol=3e-2)
    # online predictor (trainers.py:925-949): one utterance at a time reproduces the batch assessment's hypotheses
    predict = tr.construct_online_predictor()
    data = tr.net._stage(tr.ecog_subjects[-1], 'validation')
    for i in (0, 1, n - 1):
        assert predict(data['X'][i]) == [res['validation'].hypotheses[i]]
    assert predict(data['X'][:3]) == res['validation'].hypotheses[:3]
    # the checkpoint is also a TensorFlow V2 checkpoint (index + data shard): with the .npz gone, sizes, weights and EMA
    # shadows come from it and the assessment is reproduced
    assert os.path.getsize(os.path.join(ck, 'model.ckpt-60.index')) > 48
    os.remove(os.path.join(ck, 'model.ckpt-60.npz'))
    res_tf = tr.assess_saved_model()
    assert res_tf['validation'].hypotheses == res['validation'].hypotheses
    assert res_tf['validation'].accuracy == res['validation'].accuracy


def test_fit_with_bf16_staged_inputs_equals_the_fp32_staged_fit(small):
    """`input_staging='bf16'` (SURVEY.md 8 d4 'bf16 in'; reference input contract trainers.py:808-818): the training partition
    is staged once as the bf16 im2row rows of the front-end and a step gathers its operand from them -- the fit ends with the
    same weights, losses and hypotheses as the fp32-staged fit (same bits into every product)."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    out = {}
    for mode in ('fp32', 'bf16'):
        path = make_experiment(small / mode, subject_ids=(401,), epochs=6, interval=3)
        ck = str(small / mode / 'ck'); os.makedirs(ck)
        tr = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False,
                                 SN_kwargs={'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.1, 'RNN_dropout': 0.1, 'EMA_decay': 0.9,
                                            'input_staging': mode},
                                 DG_kwargs={'max_samples': 420})
        for s in tr.ecog_subjects:
            s.write_tf_records_maybe()
        a = tr.parallel_transfer_learn()
        z = np.load(os.path.join(ck, 'model.ckpt-6.npz'))
        out[mode] = (a, {k: z[k] for k in z.files})
    (a0, z0), (a1, z1) = out['fp32'], out['bf16']
    assert [l['decoder'] for l in a0['training'].losses] == [l['decoder'] for l in a1['training'].losses]
    assert a0['validation'].hypotheses == a1['validation'].hypotheses
    for k in z0:
        if 'decoder_embedding' in k or k.startswith('__'):   # (scatter-add of fp32 atomics: any order; '__adam_*' = whole-store arrays)
            np.testing.assert_allclose(z0[k], z1[k], atol=1e-5)
        else:
            assert np.array_equal(z0[k], z1[k]), k


def test_fit_with_in_graph_batch_prefetch_equals_the_fit_that_gathers_between_steps(small):
    """Round 6 (engine option prefetch_batches, Seq2SeqEngine.set_prefetch): with the partition resident in HBM the captured step
    gathers the NEXT batch into its own input buffers on a side branch under the encoder; the fit gathers itself only at the first
    step of an epoch and behind an assessment.  Several steps per epoch (N_cases 16 on 48 training utterances), two participants
    in turn, assessments every other epoch: the fit ends with the same weights, losses and hypotheses as the one that gathers
    between steps -- the same batches reached the same steps."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    out = {}
    for mode in (True, False):
        path = make_experiment(small / str(mode), subject_ids=(400, 401), epochs=6, interval=2)
        ck = str(small / str(mode) / 'ck'); os.makedirs(ck)
        tr = MultiSubjectTrainer(path, [400, 401], checkpoint_dir=ck, VERBOSE=False,
                                 SN_kwargs={'N_cases': 16, 'learning_rate': 3e-3, 'FF_dropout': 0.1, 'RNN_dropout': 0.1, 'EMA_decay': 0.9,
                                            'engine_options': {'prefetch_batches': mode}},
                                 DG_kwargs={'max_samples': 420})
        for s in tr.ecog_subjects:
            s.write_tf_records_maybe()
        a = tr.parallel_transfer_learn()
        eng = tr.net._engine
        used = [bool(w.get('prefetch')) for k, w in eng._ws.items() if isinstance(k, tuple) and len(k) == 4 and k[0] in (400, 401) and w.get('graph')]
        assert used and all(u == mode for u in used), (mode, used)
        z = np.load(os.path.join(ck, 'model.ckpt-6.npz'))
        out[mode] = (a, {k: z[k] for k in z.files})
    (a0, z0), (a1, z1) = out[True], out[False]
    assert [l['decoder'] for l in a0['training'].losses] == [l['decoder'] for l in a1['training'].losses]
    assert a0['validation'].hypotheses == a1['validation'].hypotheses
    for k in z0:
        if 'decoder_embedding' in k or k.startswith('__'):   # (scatter-add of fp32 atomics: any order; '__adam_*' = whole-store arrays)
            np.testing.assert_allclose(z0[k], z1[k], atol=1e-5)
        else:
            assert np.array_equal(z0[k], z1[k]), k


def test_sequential_transfer_and_resume(small):
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    path = make_experiment(small, subject_ids=(400, 401), epochs=2, interval=1)
    ck = str(small / 'ck'); os.makedirs(ck)
    tr = MultiSubjectTrainer(path, [400, 401], checkpoint_dir=ck, VERBOSE=False, SN_kwargs={'N_cases': 32},
                             DG_kwargs={'max_samples': 420})
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    tr.sequential_transfer_learn(pretraining_epochs=1, training_epochs=2, posttraining_epochs=1)
    assert tr.restore_epoch == 2 + 1 + 3
    for e in (2, 3, 6):
        assert os.path.exists(os.path.join(ck, 'model.ckpt-%d.index' % e))
    # during subject 401's pre-training only its own sub-network may move
    z2, z3, z6 = (np.load(os.path.join(ck, 'model.ckpt-%d.npz' % e)) for e in (2, 3, 6))
    k_shared, k_own = 'seq2seq/decoder_rnn/cell_0/kernel', 'seq2seq/subnet_401/encoder_embedding_16_24_0/weights'
    # a sub-network exists only in fits that include its subject (as in the reference's per-fit graph)
    assert k_own not in z2.files and 'seq2seq/subnet_400/encoder_embedding_16_24_0/weights' in z2.files
    assert np.array_equal(z2[k_shared], z3[k_shared])          # shared body restored and frozen during pre-training
    assert not np.array_equal(z3[k_shared], z6[k_shared])      # ... and trained afterwards
    assert not np.array_equal(z3[k_own], z6[k_own])
    a = tr.parallel_transfer_learn(RESUME=True)
    assert len(a['training'].decoder_accuracies) == tr.net.N_epochs


def test_parallel_fit_three_subjects_with_different_grids(small):
    """The README's multi-subject call (README.md:72-102 uses [400, 401]): ONE fit over three participants whose grids
    differ (4x4, 2x4, 4x4 electrodes -> per-subject conv front-ends of different widths), batches drawn round-robin.
    Every participant's front-end is trained and checkpointed, the shared body learns the (shared) sentences, and a
    resumed fit continues from the checkpoint with all three."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    path = make_experiment(small, subject_ids=(400, 401, 402), epochs=30, interval=15, grids={401: (2, 4)})
    ck = str(small / 'ck'); os.makedirs(ck)
    tr = MultiSubjectTrainer(path, [400, 401, 402], checkpoint_dir=ck, VERBOSE=False,
                             SN_kwargs={'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.1, 'EMA_decay': 0.9},
                             DG_kwargs={'max_samples': 420})
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    a = tr.parallel_transfer_learn()
    assert a['training'].losses[-1]['decoder'] < 0.6 * a['training'].losses[0]['decoder']
    assert a['validation'].decoder_word_error_rates[-1] < a['validation'].decoder_word_error_rates[0]
    z0 = np.load(os.path.join(ck, 'model.ckpt-30.npz'))
    names = {400: 'seq2seq/subnet_400/encoder_embedding_16_24_0/weights', 401: 'seq2seq/subnet_401/encoder_embedding_8_24_0/weights',
             402: 'seq2seq/subnet_402/encoder_embedding_16_24_0/weights'}
    for sid, nm in names.items():
        assert nm in z0.files and z0[nm].dtype == np.float32
        assert z0[nm].shape == (1, 12, 8 if sid == 401 else 16, 24)
    ls, ds, strides, ema = tr.recover_model_sizes()
    assert ds['401']['encoder_inputs'] == 8 and ds['400']['encoder_inputs'] == 16 and ds['402']['encoder_inputs'] == 16
    # all three front-ends moved away from their initialisation (each participant was stepped)
    eng = tr.net._engine
    init = type(eng)(eng.spec, device='cuda:0', seed=tr.net.seed)
    init.init_params(tr.net.seed)
    P0 = init.store.export_tf('p')
    for nm in names.values():
        assert np.abs(z0[nm] - P0[nm]).max() > 1e-3, nm


def test_staged_backward_graphs_match_single_graph():
    """The data-parallel step replays one hipGraph per backward stage; with a no-op exchange it must
    reproduce the single-graph step."""
    from test_gpu_parity import build, SPECS

    class FakeSync:
        world, grad_scale = 2, 1.0
        def __init__(self): self.calls = []
        def allreduce_range(self, a, b): self.calls.append((a, b))
        def allreduce_flag(self, word): pass
        def wait_flag(self): pass
        def wait(self): pass
    eng, ws, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
    eng2, ws2, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
    fs = FakeSync()
    for _ in range(3):
        eng.train_step(ws, use_graph=True)
        eng2.train_step(ws2, use_graph=True, sync=fs)
    torch.cuda.synchronize()
    np.testing.assert_allclose(eng.store.p.cpu().numpy(), eng2.store.p.cpu().numpy(), atol=1e-5)
    # every parameter element was offered for reduction exactly once per step, in backward order
    per_step = fs.calls[:len(fs.calls) // 3]
    covered = sorted(per_step)
    assert covered[0][0] == 0 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    assert covered[-1][1] == eng.store.seg_range('conv401.W')[1]


def test_staged_step_with_a_real_rccl_group():
    """Same, with the all-reduces issued through torch.distributed's "nccl" backend (= RCCL) on a one-rank group: the
    collectives run behind the side stream's graphs exactly as on a multi-GPU node, only the peers are missing."""
    import os
    import torch.distributed as dist
    from test_gpu_parity import build, SPECS
    from ecog2txt_amd.parallel import GradSync
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    except Exception as e:                                   # no usable network interface on the box
        pytest.skip('cannot create an RCCL group here: %r' % (e,))
    try:
        class OneRankAsMany(GradSync):
            """world is reported as 2 so that the engine takes the data-parallel path; the sum over the one real rank
            leaves the gradients unchanged, so the scale stays 1."""
            def __init__(self, g):
                super().__init__(g)
                self.world = 2
            def allreduce_range(self, a, b):
                if b > a:
                    w = dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, async_op=True)
                    self.pending.append(w)
                    self.pending_ranges.append((w, a, b))        # the optimiser follows the exchange range by range
            grad_scale = 1.0
        eng, ws, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
        eng2, ws2, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
        sync = OneRankAsMany(eng2.store.g)
        for _ in range(4):
            eng.train_step(ws, use_graph=True)
            eng2.train_step(ws2, use_graph=True, sync=sync)
        torch.cuda.synchronize()
        assert int(eng2.sync_err[0].item()) == 0
        np.testing.assert_allclose(eng.store.p.cpu().numpy(), eng2.store.p.cpu().numpy(), atol=1e-5)
        assert eng.losses(ws)['total'] == pytest.approx(eng2.losses(ws2)['total'], rel=1e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('one_graph', [True, False], ids=['one_graph', 'graph_per_stage'])
def test_data_parallel_step_with_direct_rccl_through_the_c_abi(one_graph):
    """The product's exchange: librccl called directly through e2t_comm_* (no torch.distributed anywhere in the step) on a
    one-rank communicator -- collectives on the communicator's own stream, ordered by events behind the stream that completes
    their ranges, the optimiser following the exchange.  Both schedules: the step as ONE graph with the collectives as nodes
    (option dp_one_graph=True) and one graph per backward stage with the collectives issued between them (the default until the
    one-graph schedule has run with more than one RCCL rank, and the fallback all ranks take together).  One real rank reported as two takes the engine down the data-parallel path; the sum over one rank leaves the
    gradients unchanged, so the result must equal the single-graph step.  Also: global-count loss normalisation equals local
    normalisation when the counts agree."""
    from test_gpu_parity import build, SPECS
    from ecog2txt_amd.parallel import RcclSync
    eng, ws, _, _, batch = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
    eng2, ws2, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9, options={'dp_one_graph': one_graph})
    try:
        sync = RcclSync(eng2.store.g, 0, 1, RcclSync.unique_id(), 0, sum_of_global_means=True)
    except RuntimeError as e:
        pytest.skip('cannot create an RCCL communicator here: %r' % (e,))
    try:
        sync.world = 2                                       # (the communicator itself has one rank)
        ntok, nval = eng2.local_counts(batch['decoder_targets'], batch['encoder_targets'])
        eng2.set_global_counts(ws2, ntok, nval)
        for _ in range(4):
            eng.train_step(ws, use_graph=True)
            eng2.train_step(ws2, use_graph=True, sync=sync)
        torch.cuda.synchronize()
        assert int(eng2.sync_err[0].item()) == 0
        assert any(k[0] == 'train_dp' and k[6] == one_graph for k in ws2['graph'] if isinstance(k, tuple))
        np.testing.assert_allclose(eng.store.p.cpu().numpy(), eng2.store.p.cpu().numpy(), atol=1e-5)
        assert eng.losses(ws)['total'] == pytest.approx(eng2.losses(ws2)['total'], rel=1e-5)
        # the small host-side exchanges of the sharded assessment, and the parameter broadcast
        a = np.arange(12, dtype=np.int32).reshape(3, 4)
        assert np.array_equal(sync.allreduce_numpy(a), a)
        f = np.linspace(0, 1, 7).astype(np.float32)
        assert np.array_equal(sync.allreduce_numpy(f), f)
        before = eng2.store.p.clone()
        sync.broadcast_([eng2.store.p])
        torch.cuda.synchronize()
        assert torch.equal(before, eng2.store.p)
    finally:
        sync.close()


def test_optimiser_skips_the_update_after_an_in_kernel_timeout():
    """a step whose persistent recurrence raised the error word must not reach the weights -- Adam, EMA and the
    step counter are no-ops on the device while the word is set -- and losses() / check_sync() raise and clear it."""
    from test_gpu_parity import build, SPECS
    eng, ws, *_ = build(SPECS['small_dropout'], 19, 26, 6, seed=9)
    eng.train_step(ws, use_graph=True)
    torch.cuda.synchronize()
    p1, e1, t1 = eng.store.p.clone(), eng.store.ema.clone(), int(eng.step_t.item())
    eng.sync_err[0] = 1                                      # as the bounded wait of a persistent kernel would
    eng.train_step(ws, use_graph=True)
    eng.train_step(ws, use_graph=False)
    torch.cuda.synchronize()
    assert torch.equal(eng.store.p, p1) and torch.equal(eng.store.ema, e1) and int(eng.step_t.item()) == t1
    with pytest.raises(RuntimeError, match='timed out'):
        eng.losses(ws)
    assert int(eng.sync_err[0].item()) == 0
    eng.train_step(ws, use_graph=True)                       # and training goes on
    torch.cuda.synchronize()
    assert not torch.equal(eng.store.p, p1) and int(eng.step_t.item()) == t1 + 1


@pytest.mark.parametrize('dp', [False, True], ids=['single', 'data_parallel_one_graph'])
def test_error_word_raised_inside_the_step_leaves_every_range_untouched(dp):
    """the error word is raised INSIDE a captured step (non-finite gate gradients make the persistent BPTT raise code 7
    -- at the earliest by the top layer, i.e. after the head's gradients are complete) and the early optimiser update on the side
    stream must still see it: masters, Adam state, EMA shadows and the step counter of EVERY range stay as they were -- also in
    the data-parallel graph, where the ranks agree on the word (maximum) before any update reads it."""
    from test_gpu_parity import build, SPECS
    from ecog2txt_amd.parallel import RcclSync
    kw = dict(SPECS['cfg2_widths'], enc_rnn=[400, 400])         # persistent recurrences (H = 400), two layers: an early update exists
    eng, ws, _, _, batch = build(kw, 64, 40, 5, seed=3, options={'dp_one_graph': True})
    assert eng.persistent_bwd and eng.enc[0].persistent_bwd_ok(64, eng.num_cus)
    sync = None
    if dp:
        try:
            sync = RcclSync(eng.store.g, 0, 1, RcclSync.unique_id(), 0, sum_of_global_means=True)
        except RuntimeError as e:
            pytest.skip('cannot create an RCCL communicator here: %r' % (e,))
        sync.world = 2
        ntok, nval = eng.local_counts(batch['decoder_targets'], batch['encoder_targets'])
        eng.set_global_counts(ws, ntok, nval)
    try:
        for _ in range(2):
            eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize()
        assert int(eng.sync_err[0].item()) == 0
        st = eng.store
        before = [t.clone() for t in (st.p, st.m, st.v, st.ema)]
        t1 = int(eng.step_t.item())
        good = ws['auxT'].clone()
        ws['auxT'][3, :, 0] = float('inf')                       # -> non-finite auxiliary gradient -> dY of layer 0 -> its BPTT raises
        eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize()
        assert int(eng.sync_err[0].item()) != 0
        for a, b in zip(before, (st.p, st.m, st.v, st.ema)):
            assert torch.equal(a, b)
        assert int(eng.step_t.item()) == t1
        with pytest.raises(RuntimeError, match='not updated'):
            eng.check_sync(ws)
        ws['auxT'].copy_(good)
        eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize()
        assert int(eng.sync_err[0].item()) == 0 and int(eng.step_t.item()) == t1 + 1 and not torch.equal(before[0], st.p)
    finally:
        if sync is not None:
            sync.close()


def test_forward_after_a_captured_step_sees_the_updated_weights():
    """The captured train step re-packs the bf16 operand images at its START (next to the weight-free front-end), so
    after it returns they are one update behind the fp32 masters: a forward pass that follows must re-pack first."""
    from test_gpu_parity import build, SPECS
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    eng, ws, ospec, P, batch = build(SPECS['small_dropout'], 19, 26, 6, seed=3)
    for _ in range(3):
        eng.train_step(ws, use_graph=True)
    eng.forward(ws, train=False)
    torch.cuda.synchronize()
    got = eng.losses(ws)
    ref = Seq2SeqEngine(NetSpec(**SPECS['small_dropout']), device='cuda:0', seed=11)
    ref.store.p.copy_(eng.store.p); ref.store.ema.copy_(eng.store.ema)
    ref.pack('p')
    ws2 = ref.workspace(401, 19, 26, 6)
    ref.set_batch(ws2, batch)
    ref.forward(ws2, train=False)
    torch.cuda.synchronize()
    want = ref.losses(ws2)
    assert got['total'] == pytest.approx(want['total'], rel=1e-6)
    # and the weights did move
    assert got['total'] < 0.999 * eng_initial_loss(SPECS['small_dropout'], batch)


def eng_initial_loss(kw, batch):
    from test_gpu_parity import build
    eng, ws, *_ = build(kw, 19, 26, 6, seed=3)
    eng.forward(ws, train=False)
    torch.cuda.synchronize()
    return eng.losses(ws)['total']


def test_fit_with_two_auxiliary_target_keys(small):
    """Every 'encoder_<k>_targets' data key is an auxiliary head on encoder layer k (trainers.py:94-102, 786-799): two audio
    heads (layers 0 and 1), both losses reported and falling, both heads' variables in the checkpoint under the reference's
    names, restore + assess intact."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    path = make_experiment(small, subject_ids=(401,), epochs=30, interval=15, extra_aux=True)
    ck = str(small / 'ck2'); os.makedirs(ck)
    tr = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False,
                             SN_kwargs={'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.1, 'EMA_decay': 0.9},
                             DG_kwargs={'max_samples': 420})
    dms = tr.ecog_subjects[-1].data_manifests
    assert dms['encoder_0_targets'].penalty_scale == 0.25 and dms['encoder_1_targets'].penalty_scale == 0.5
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    a = tr.parallel_transfer_learn()
    spec = tr.net._engine.spec
    assert spec.aux_layer == 0 and spec.aux_hidden == [12] and spec.aux_scale == 0.25
    assert spec.aux_extra == [dict(layer=1, hidden=[24], dim=5, dist='Gaussian', scale=0.5)]
    lo = a['training'].losses
    assert 'aux' in lo[0] and 'aux_x0' in lo[0]
    assert lo[-1]['aux'] < lo[0]['aux'] and lo[-1]['aux_x0'] < lo[0]['aux_x0'] and lo[-1]['decoder'] < lo[0]['decoder']
    from ecog2txt_amd import tf_checkpoint
    names = {k: list(v) for k, v in tf_checkpoint.list_variables(os.path.join(ck, 'model.ckpt-30'))}
    assert names['seq2seq/encoder_0_projection_64_12_0/weights'] == [64, 12]
    assert names['seq2seq/encoder_1_projection_64_24_0/weights'] == [64, 24]
    assert names['seq2seq/encoder_0_projection_12_5_1/weights'] == [5, 12] and names['seq2seq/encoder_1_projection_24_5_1/weights'] == [5, 24]
    res = tr.assess_saved_model()
    assert np.isfinite(res['validation'].word_error_rate)


def test_fit_with_a_stack_of_conv_layers(small):
    """layer_sizes['encoder_embedding'] with two entries = two strided conv layers whose strides (4 x 3) multiply to the
    decimation factor 12 (trainers.py:406-407, 535-541; the split is the `encoder_strides` argument): the fit learns, the
    checkpoint holds both layers under the reference's names with rank-4 weights whose width is the stride, and a FRESH trainer
    recovers sizes and strides from the checkpoint alone and reproduces the assessment."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    path = make_experiment(small, subject_ids=(401,), epochs=40, interval=20, embedding=(20, 24))
    ck = str(small / 'ck3'); os.makedirs(ck)
    kw = dict(checkpoint_dir=ck, VERBOSE=False, DG_kwargs={'max_samples': 420})
    sn = {'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.05, 'RNN_dropout': 0.1, 'EMA_decay': 0.9}
    tr = MultiSubjectTrainer(path, [401], SN_kwargs=dict(sn, encoder_strides=[4, 3]), **kw)
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    a = tr.parallel_transfer_learn()
    assert tr.net._engine.spec.conv_pre == [dict(out=20, stride=4)] and tr.net._engine.spec.enc_embed == 24
    lo = a['training'].losses
    assert lo[-1]['decoder'] < 0.6 * lo[0]['decoder'] and lo[-1]['aux'] < lo[0]['aux']
    z = np.load(os.path.join(ck, 'model.ckpt-40.npz'))
    assert z['seq2seq/subnet_401/encoder_embedding_16_20_0/weights'].shape == (1, 4, 16, 20)
    assert z['seq2seq/subnet_401/encoder_embedding_20_24_1/weights'].shape == (1, 3, 20, 24)
    assert z['seq2seq/subnet_401/encoder_embedding_20_24_1/biases'].shape == (24,)
    res = tr.assess_saved_model()
    tr2 = MultiSubjectTrainer(path, [401], SN_kwargs=sn, **kw)            # no encoder_strides: they come from the checkpoint
    ls, ds, strides, ema = tr2.recover_model_sizes()
    assert ls['encoder_embedding'] == [20, 24] and strides['401'] == [4, 3] and ds['401']['encoder_inputs'] == 16
    res2 = tr2.assess_saved_model()
    assert tr2.net.encoder_strides == [4, 3] and tr2.ecog_subjects[-1].decimation_factor == 12
    assert res2['validation'].hypotheses == res['validation'].hypotheses and res2['validation'].accuracy == res['validation'].accuracy
    sal = tr2.get_saliencies('decoder_saliency_map')
    assert sal.shape == (16,) and np.isfinite(sal).all() and sal.max() > 0
    act = tr2.get_internal_activations()
    assert act['convolved_inputs'].shape[2] == 24


def test_fit_on_audio_targets_made_by_speech_features(small):
    """SURVEY 8 f4 on the GPU: the auxiliary targets of a fit are MFCCs that `speech_features.mfcc_features` computes from
    (synthetic) waveforms through the reference's hooks -- `_get_wav_data` -> `_get_MFCC_features(index, winstep)`
    (ecog2txt/data_generators.py:328-380) -> `audio_sequence` in the records -> `encoder_1_targets` of the HIP train step.
    The records hold exactly what the feature code returns, the fit consumes them, the auxiliary loss falls."""
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    from ecog2txt_amd import tfrecord
    from ecog2txt_amd.speech_features import mfcc_features
    path = make_experiment(small, subject_ids=(401,), epochs=30, interval=15, generator='SyntheticWaveformDataGenerator')
    ck = str(small / 'ckw'); os.makedirs(ck)
    tr = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False,
                             SN_kwargs={'N_cases': 32, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.1, 'EMA_decay': 0.9},
                             DG_kwargs={'max_samples': 420})
    subj = tr.ecog_subjects[-1]
    dg = subj.data_generator
    assert type(dg).__name__ == 'SyntheticWaveformDataGenerator' and dg.num_MFCC_features == 5
    for s in tr.ecog_subjects:
        s.write_tf_records_maybe()
    # the records carry the feature code's output for the trial's waveform (one frame per ECoG sample)
    block = sorted(subj.block_ids['training'])[0]
    payloads = list(tfrecord.tf_record_iterator(dg.tf_record_partial_path.format(block)))
    ex = tfrecord.decode_example(payloads[3])
    rate, wav = dg._get_wav_data((block, 3))
    want = mfcc_features(wav, rate, dg.mfcc_winlen, 1.0 / dg.sampling_rate, dg.num_mel_features, dg.num_cepstral_coeffs, False, False, 512)
    got = np.asarray(ex['audio_sequence'], np.float32).reshape(-1, 5)
    T = np.asarray(ex['ecog_sequence']).size // 16
    # python_speech_features' framing: 1 + ceil((samples - window) / step) frames -- (window / step - 1) fewer than ECoG samples;
    # the generator holds the last frame over the tail (the reference leaves the alignment to the subclass hook)
    nwin, nstep = int(round(dg.mfcc_winlen * rate)), int(round(rate / dg.sampling_rate))
    assert got.shape[0] == T and want.shape[0] == 1 + int(np.ceil((wav.shape[0] - nwin) / nstep)) and 0 <= T - want.shape[0] <= nwin // nstep
    n = min(T, want.shape[0])
    np.testing.assert_allclose(got[:n], want[:n].astype(np.float32), rtol=1e-6, atol=1e-6)
    assert np.abs(got[:, 0]).min() > 1.0            # c0 = log frame energy of a non-silent waveform
    a = tr.parallel_transfer_learn()
    lo = a['training'].losses
    assert 'aux' in lo[0] and np.isfinite(lo[-1]['aux'])
    assert lo[-1]['aux'] < 0.5 * lo[0]['aux'], (lo[0], lo[-1])
    assert lo[-1]['decoder'] < lo[0]['decoder']
    res = tr.assess_saved_model()
    assert np.isfinite(res['validation'].word_error_rate)
