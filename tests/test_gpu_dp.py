"""Two-rank data-parallel fit on ONE GPU (both processes share cuda:0, gradients exchanged with torch.distributed "gloo"):
the whole data-parallel control flow of SequenceNetwork.fit on the real kernels -- global batches cut per rank, a rank with an
EMPTY slice in the last step (n % (world * N_cases) != 0), losses normalised by the global counts, per-stage exchange, sharded
assessment, rank-0 checkpoint -- against a single-process fit over the same global batches.
Dropout is off (the ranks key their masks differently) and the launch-per-step recurrences are used (two processes cannot
both keep a persistent kernel's workgroups co-resident on one GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, root, ckdir, n_cases, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      E2T_COMM='torch')
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    ECoGDataGenerator.text_dir = root
    SyntheticSpeechDataGenerator.num_sentences = 6
    SyntheticSpeechDataGenerator.trials_per_block = 24
    SyntheticSpeechDataGenerator.max_words = 5
    tr = MultiSubjectTrainer(os.path.join(root, 'experiment.yaml'), [401], checkpoint_dir=ckdir, VERBOSE=False,
                             SN_kwargs={'N_cases': n_cases, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.0, 'EMA_decay': 0.9,
                                        'engine_options': {'persistent': '0'}},
                             DG_kwargs={'max_samples': 420})
    a = tr.parallel_transfer_learn()
    out = dict(rank=rank, losses=a['training'].losses, wer=a['validation'].decoder_word_error_rates.tolist(),
               acc=a['validation'].decoder_accuracies.tolist(), hyp=a['validation'].hypotheses)
    q.put(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, port, root, ckdir, n_cases):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, root, ckdir, n_cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r['rank'])


def test_two_rank_fit_on_one_gpu_equals_the_single_process_fit(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from experiment_fixture import make_experiment
    from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
    from ecog2txt_amd.subjects import ECoGSubject
    from ecog2txt_amd.manifests import load_manifest
    root = str(tmp_path)
    path = make_experiment(root, subject_ids=(401,), epochs=4, interval=2)
    # records once, by the parent (the workers find them on disk)
    ECoGDataGenerator.text_dir = root
    SyntheticSpeechDataGenerator.num_sentences, SyntheticSpeechDataGenerator.trials_per_block, SyntheticSpeechDataGenerator.max_words = 6, 24, 5
    sub = ECoGSubject(load_manifest(path)[401], 401, _DG_kwargs={'max_samples': 420})
    sub.write_tf_records_maybe()
    ck1, ck2 = os.path.join(root, 'ck1'), os.path.join(root, 'ck2')
    os.makedirs(ck1); os.makedirs(ck2)
    port = 32500 + (os.getpid() % 1500)
    one = _run(1, port, root, ck1, 32)[0]
    two = _run(2, port + 1, root, ck2, 16)               # 72 training utterances: global batches of 32, 32, 8 -> rank 1's last slice is EMPTY
    z1, z2 = np.load(os.path.join(ck1, 'model.ckpt-4.npz')), np.load(os.path.join(ck2, 'model.ckpt-4.npz'))
    assert set(z1.files) == set(z2.files)
    nsteps = 4 * 3
    for k in z1.files:
        if k.startswith('__adam') or k == '__step':
            continue
        # same global batches, same normalisation: the trajectories agree up to the order of fp32 sums (which Adam can amplify
        # to a fraction of a step on near-zero gradients)
        assert np.abs(z1[k] - z2[k]).max() < nsteps * 3e-3 * 0.35, k
    assert int(z1['__step'][0]) == int(z2['__step'][0]) == nsteps          # every rank ran every step
    # both ranks report the same (summed) losses and the same assessments; they match the single-process run
    assert two[0]['losses'] == two[1]['losses'] and two[0]['hyp'] == two[1]['hyp'] and two[0]['wer'] == two[1]['wer']
    for a, b in zip(one['losses'], two[0]['losses']):
        assert abs(a['decoder'] - b['decoder']) < 5e-2 * max(1.0, abs(a['decoder'])), (a, b)
    assert len(two[0]['hyp']) == len(one['hyp'])
    assert abs(two[0]['acc'][-1] - one['acc'][-1]) < 0.1


def test_bench_under_the_launcher_with_two_ranks_on_one_gpu():
    """bench.py exactly as the driver starts it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`),
    with both ranks mapped onto the one GPU of this box (E2T_BENCH_BACKEND=gloo, --engine-option persistent=0: launch-per-step recurrences): rank 0 prints
    ONE JSON line for the whole job, the other rank waits at the closing barrier while rank 0 measures the rooflines, exit 0."""
    import json, socket, subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, E2T_COMM='torch', E2T_BENCH_BACKEND='gloo')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--engine-option', 'persistent=0'],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['config']['global_batch'] == 512 and d['config']['parallelism'] == 'dp2'
    assert d['scaling'] == 'weak' and d['roofline'] is not None and 'cpu_baseline' not in d and np.isfinite(d['final_loss'])
    assert abs(d['value'] - 512 * 1e3 / d['ms_per_step']) < 1e-2 * d['value']
