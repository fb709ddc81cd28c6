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


def test_staged_backward_graphs_match_single_graph():
    """The data-parallel step replays one hipGraph per backward stage; with a no-op exchange it must
    reproduce the single-graph step."""
    from test_gpu_parity import build, SPECS

    class FakeSync:
        world, grad_scale = 2, 1.0
        def __init__(self): self.calls = []
        def allreduce_range(self, a, b): self.calls.append((a, b))
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
