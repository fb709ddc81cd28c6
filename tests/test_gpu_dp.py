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


def _worker(rank, world, port, root, ckdir, n_cases, q, rccl=False, engine_options=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    # rccl: one GPU per rank, gradients exchanged by librccl through the C ABI (the product path; no torch.distributed group)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank) if rccl else '0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      E2T_COMM='rccl' if rccl else 'torch', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    if world > 1 and not rccl:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
    from ecog2txt_amd.trainers import MultiSubjectTrainer
    ECoGDataGenerator.text_dir = root
    SyntheticSpeechDataGenerator.num_sentences = 6
    SyntheticSpeechDataGenerator.trials_per_block = 24
    SyntheticSpeechDataGenerator.max_words = 5
    tr = MultiSubjectTrainer(os.path.join(root, 'experiment.yaml'), [401], checkpoint_dir=ckdir, VERBOSE=False,
                             SN_kwargs={'N_cases': n_cases, 'learning_rate': 3e-3, 'FF_dropout': 0.0, 'RNN_dropout': 0.0, 'EMA_decay': 0.9,
                                        'engine_options': engine_options if engine_options is not None else {'persistent': '0'}},
                             DG_kwargs={'max_samples': 420})
    a = tr.parallel_transfer_learn()
    out = dict(rank=rank, losses=a['training'].losses, wer=a['validation'].decoder_word_error_rates.tolist(),
               acc=a['validation'].decoder_accuracies.tolist(), hyp=a['validation'].hypotheses)
    out['comm_ranks'] = int(tr.net._sync.lib.e2t_comm_size(tr.net._sync.comm)) if (rccl and world > 1) else world
    q.put(out)
    if world > 1 and not rccl:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1 and rccl:
        tr.net._sync.barrier()
        tr.net._sync.close()


def _run(world, port, root, ckdir, n_cases, **kw):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, root, ckdir, n_cases, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:                  # (a rank that hangs -- in a collective, say -- must not outlive the test)
            if p.is_alive():
                p.kill()
                p.join(timeout=30)
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


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (live the day the driver has a node)')


@needs_two_gpus
@pytest.mark.parametrize('one_graph', [False, True], ids=['graph_per_stage', 'one_graph'])
def test_two_rank_rccl_fit_on_two_gpus_equals_the_single_process_fit(tmp_path, one_graph):
    """The product's data-parallel fit with REAL peers (missing 3): two processes, one GPU each, librccl over xGMI through
    the C ABI, persistent recurrences, both step schedules (one graph per backward stage with eager collectives -- the default -- and
    the step as one graph with the collectives as nodes) -- against the single-process fit over the same global batches.  Same
    tolerances as the two-ranks-on-one-GPU gloo test below (the sum of two ranks' fp32 gradients is associated differently from
    one rank's sum over the whole batch; the embedding gradient is a scatter-add of fp32 atomics)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from experiment_fixture import make_experiment
    from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
    from ecog2txt_amd.subjects import ECoGSubject
    from ecog2txt_amd.manifests import load_manifest
    root = str(tmp_path)
    path = make_experiment(root, subject_ids=(401,), epochs=4, interval=2)
    ECoGDataGenerator.text_dir = root
    SyntheticSpeechDataGenerator.num_sentences, SyntheticSpeechDataGenerator.trials_per_block, SyntheticSpeechDataGenerator.max_words = 6, 24, 5
    sub = ECoGSubject(load_manifest(path)[401], 401, _DG_kwargs={'max_samples': 420})
    sub.write_tf_records_maybe()
    ck1, ck2 = os.path.join(root, 'ck1'), os.path.join(root, 'ck2')
    os.makedirs(ck1); os.makedirs(ck2)
    port = 34500 + (os.getpid() % 1500)
    opts = {'dp_one_graph': one_graph}
    one = _run(1, port, root, ck1, 32, rccl=True, engine_options=opts)[0]
    two = _run(2, port + 2, root, ck2, 16, rccl=True, engine_options=opts)
    assert [r['comm_ranks'] for r in two] == [2, 2]
    z1, z2 = np.load(os.path.join(ck1, 'model.ckpt-4.npz')), np.load(os.path.join(ck2, 'model.ckpt-4.npz'))
    nsteps = 4 * 3
    for k in z1.files:
        if k.startswith('__adam') or k == '__step':
            continue
        assert np.abs(z1[k] - z2[k]).max() < nsteps * 3e-3 * 0.35, k
    assert int(z1['__step'][0]) == int(z2['__step'][0]) == nsteps
    assert two[0]['losses'] == two[1]['losses'] and two[0]['hyp'] == two[1]['hyp'] and two[0]['wer'] == two[1]['wer']
    for a, b in zip(one['losses'], two[0]['losses']):
        assert abs(a['decoder'] - b['decoder']) < 5e-2 * max(1.0, abs(a['decoder'])), (a, b)
    assert abs(two[0]['acc'][-1] - one['acc'][-1]) < 0.1


def _dp_step_worker(rank, world, port, one_graph, q):
    """cfg2's real step with a real peer: 6 data-parallel steps; returns a checksum of the parameters and the step time."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import time
    import bench
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    from ecog2txt_amd import parallel
    torch.cuda.set_device(rank)
    kw, B, T, L = bench.CONFIGS['cfg2']
    eng = Seq2SeqEngine(NetSpec(**kw), device='cuda:%d' % rank, seed=3 + rank, options={'dp_one_graph': one_graph})
    eng.init_params(seed=0)
    sync = parallel.make_sync(eng.store.g)
    sync.broadcast_([eng.store.p, eng.store.ema])
    eng.pack('p')
    ws = eng.workspace(401, B, T, L)
    batch = bench.synth_batch(kw, B, T, L, seed=5 + rank)
    eng.set_batch(ws, batch)
    cnt = sync.allreduce_numpy(np.array(eng.local_counts(batch['decoder_targets'], batch['encoder_targets']), np.int64))
    eng.set_global_counts(ws, int(cnt[0]), int(cnt[1]))
    with eng.on_step_stream():
        for _ in range(3):
            eng.train_step(ws, sync=sync)
        torch.cuda.synchronize(); sync.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.train_step(ws, sync=sync)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    one = any(isinstance(k, tuple) and k[0] == 'train_dp' and k[6] for k in ws['graph'])
    q.put(dict(rank=rank, ms=1e3 * dt, err=int(eng.sync_err[0].item()), step=int(eng.step_t.item()), one_graph=one,
               p=eng.store.p.double().sum().item(), pabs=eng.store.p.abs().double().sum().item(), loss=eng.losses(ws)['total'],
               ranks=int(sync.lib.e2t_comm_size(sync.comm))))
    sync.barrier()
    sync.close()


@needs_two_gpus
@pytest.mark.parametrize('one_graph', [False, True], ids=['graph_per_stage', 'one_graph'])
def test_cfg2_data_parallel_step_with_a_real_peer(one_graph):
    """cfg2's captured step (B = 256 per GPU, persistent recurrences, all-reduces behind the backward stages) on two GPUs: both
    ranks apply every step, raise no error word and hold bit-identical replicas afterwards (same summed gradients, same optimiser)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_dp_step_worker, args=(r, 2, port, one_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r['rank'])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
                p.join(timeout=30)
    a, b = res
    assert a['ranks'] == b['ranks'] == 2 and a['err'] == b['err'] == 0 and a['step'] == b['step'] == 13
    assert a['p'] == b['p'] and a['pabs'] == b['pabs']                      # replicas: bit-identical parameters
    assert np.isfinite(a['loss']) and np.isfinite(b['loss'])
    print('\ncfg2 data-parallel step on two GPUs (%s): %.3f / %.3f ms per step' % ('one graph' if a['one_graph'] else 'graph per stage', a['ms'], b['ms']))


@pytest.mark.parametrize('launcher', [True, False], ids=['torch.distributed.run', 'self_spawned'])
def test_bench_with_two_ranks_on_one_gpu(launcher):
    """bench.py exactly as the driver starts it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)
    and as plain `python bench.py --gpus N` (it then starts its ranks itself), with both ranks mapped onto the one GPU of this box
    (E2T_BENCH_BACKEND=gloo, --engine-option persistent=0: launch-per-step recurrences): rank 0 prints ONE JSON line for the whole
    job, the other rank waits at the closing barrier while rank 0 measures the rooflines, exit 0."""
    import json, socket, subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, E2T_COMM='torch', E2T_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    pre = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port)] if launcher else [sys.executable]
    out = subprocess.run(pre + [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--engine-option', 'persistent=0'],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['config']['global_batch'] == 512 and d['config']['parallelism'] == 'dp2'
    assert d['config']['communicator_ranks'] == 2
    assert d['scaling'] == 'weak' and d['roofline'] is not None and 'cpu_baseline' not in d and np.isfinite(d['final_loss'])
    assert abs(d['value'] - 512 * 1e3 / d['ms_per_step']) < 1e-2 * d['value']


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 8` on a box with fewer GPUs must fail, not report one GPU eight times (missing 3)."""
    import subprocess
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'E2T_BENCH_BACKEND')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and 'GPU(s) are visible' in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith('{')]
