"""world_size-2 CPU (gloo) test of the data-parallel gradient exchange.

Each rank computes ORACLE gradients on its shard (test infrastructure standing in for the HIP
backward, which needs a GPU), writes them into a flat buffer laid out by the product's ParamStore,
all-reduces bucket by bucket through ecog2txt_amd.parallel.GradSync, and the rank-averaged
result must equal the oracle gradient of the whole batch (equal token counts per shard)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import seq2seq as O
    from helpers import tiny_spec, make_batch
    from ecog2txt_amd.engine import ParamStore, NetSpec
    from ecog2txt_amd.parallel import GradSync, shard_range, broadcast_flat
    ospec = tiny_spec()
    spec = NetSpec(**{k: getattr(ospec, k) for k in NetSpec.__dataclass_fields__})
    P = O.init_params(ospec, seed=3)
    batch = make_batch(ospec, B=8, T=11, L=5, seed=1, ragged=False)     # equal lengths => equal token counts
    lo, hi = shard_range(8, rank, world)
    shard = {k: (v[lo:hi] if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
    _, cache = O.forward(P, ospec, shard)
    G = O.backward(P, cache)
    store = ParamStore(spec, 'cpu')
    # params: rank 1 starts from garbage and must receive rank 0's values
    if rank == 0:
        store.import_tf(P)
    else:
        store.p.normal_()
    broadcast_flat([store.p])
    got_p = store.export_tf('p')
    store.import_tf(G, bufs=('g',))
    sync = GradSync(store.g)
    # bucket ranges in backward order, exactly as the engine would issue them
    names = store.order
    for nm in names:
        a, b = store.seg_range(nm)
        sync.allreduce_range(a, b)
    # the ranks' "this step is invalid" words are agreed on with the gradients: rank 1 raises it, every rank must see it -- WITH its
    # code, and it must stay what it is however many steps pass before the host looks (the word is reduced in place at every step
    # and only cleared by check_sync once per epoch: a sum would double it per step and wrap an int32 to 0 after 32 steps)
    flag = torch.tensor([7 if rank == 1 else 0], dtype=torch.int32)
    for _ in range(40):
        sync.allreduce_flag(flag)
        sync.wait_flag()
        assert int(flag.item()) == 7, int(flag.item())
    sync.wait()
    store.g.mul_(sync.grad_scale)
    out = store.export_tf('g')
    if rank == 0:
        _, cache_all = O.forward(P, ospec, batch)
        Gall = O.backward(P, cache_all)
        err = max(np.abs(out[k] - Gall[k]).max() / (np.abs(Gall[k]).max() + 1e-12) for k in Gall)
        perr = max(np.abs(got_p[k] - P[k].astype(np.float32)).max() for k in P)
        q.put((float(err), float(perr), sync.world))
    else:
        perr = max(np.abs(got_p[k] - P[k].astype(np.float32)).max() for k in P)
        q.put((0.0, float(perr), sync.world))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for err, perr, world in res:
        assert world == 2
        assert perr == 0.0                 # broadcast delivered rank 0's parameters bit-exactly
        assert err < 1e-6                  # fp32 flat buffers vs fp64 oracle


def test_shard_range_covers_everything():
    from ecog2txt_amd.parallel import shard_range
    for n in (1, 7, 8, 256, 257):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


# ---------------------------------------------------------------------------------------------------------------------
# uneven shards: n % (world * B) != 0 -- every rank runs the same number of steps, the empty / short slices
# are padding utterances, and with the losses normalised by the GLOBAL counts the plain SUM of the ranks' gradients is
# the gradient of the global mean loss although the ranks hold different token counts
# ---------------------------------------------------------------------------------------------------------------------
def test_global_batches_give_every_rank_the_same_number_of_steps():
    from ecog2txt_amd.parallel import global_batches, rank_slice
    for n, world, B in ((257, 2, 128), (9, 2, 2), (1, 8, 4), (256, 8, 32), (300, 3, 7)):
        rng = np.random.default_rng(5)
        gb = global_batches(n, B, world, rng)
        assert len(gb) == -(-n // (B * world))
        seen = []
        for g in gb:
            parts = [rank_slice(g, B, r) for r in range(world)]
            assert all(len(p) <= B for p in parts)
            assert sum(len(p) for p in parts) == len(g)
            seen += [i for p in parts for i in p]
        assert sorted(seen) == list(range(n))                # disjoint cover
        # the same with the global batch dealt out by length: still a disjoint cover with at most B per rank, every rank's
        # utterances are as long in total as any other's (to within one utterance), and a short last batch is SPREAD over the ranks
        lengths = np.random.default_rng(n).integers(1, 400, size=n)
        seen = []
        for g in gb:
            parts = [rank_slice(g, B, r, world, lengths) for r in range(world)]
            assert all(len(p) <= B for p in parts) and sum(len(p) for p in parts) == len(g)
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            tot = [int(lengths[p].sum()) for p in parts]
            assert max(tot) - min(tot) <= int(lengths[g].max())
            seen += [i for p in parts for i in p]
        assert sorted(seen) == list(range(n))


def _uneven_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import seq2seq as O
    from helpers import tiny_spec, make_batch
    from ecog2txt_amd.engine import ParamStore, NetSpec
    from ecog2txt_amd.parallel import GradSync, global_batches, rank_slice
    ospec = tiny_spec()
    spec = NetSpec(**{k: getattr(ospec, k) for k in NetSpec.__dataclass_fields__})
    P = O.init_params(ospec, seed=3)
    n, B = 9, 2                                              # global batches of 4, 4 and 1: rank 1's last slice is empty
    data = make_batch(ospec, B=n, T=11, L=5, seed=1, ragged=True)       # ragged: unequal token counts per rank
    store = ParamStore(spec, 'cpu')
    sync = GradSync(store.g, sum_of_global_means=True)
    assert sync.grad_scale == 1.0
    errs = []
    steps = 0
    for g in global_batches(n, B, world, np.random.default_rng(0)):
        mine = rank_slice(g, B, rank)
        # this rank's batch: its slice, padded to B rows with zero-length utterances (what e2t_gather_rows_u32 makes of idx -1)
        shard = {}
        for k, v in data.items():
            if isinstance(v, np.ndarray):
                a = np.zeros((B,) + v.shape[1:], v.dtype)
                a[:len(mine)] = v[mine]
                shard[k] = a
            else:
                shard[k] = v
        Yg = data['decoder_targets'][g]
        Ag = data['encoder_targets'][g]
        counts = (int((Yg != 0).sum()), int((-(-(np.abs(Ag).max(axis=2) > 0).sum(1) // ospec.decimation)).sum()))
        _, cache = O.forward(P, ospec, shard, counts=counts)
        G = O.backward(P, cache)
        store.g.zero_()
        store.import_tf(G, bufs=('g',))
        for nm in store.order:                               # every rank issues every collective, also with an empty slice
            a, b = store.seg_range(nm)
            sync.allreduce_range(a, b)
        sync.wait()
        steps += 1
        out = store.export_tf('g')
        whole = {k: (v[g] if isinstance(v, np.ndarray) else v) for k, v in data.items()}
        _, cache_all = O.forward(P, ospec, whole)
        Gall = O.backward(P, cache_all)
        errs.append(max(np.abs(out[k] - Gall[k]).max() / (np.abs(Gall[k]).max() + 1e-12) for k in Gall))
    # sharded assessment: every rank "decodes" its slice, the sum over ranks is the whole partition on every rank
    hyp = np.zeros((n, 3), np.int32)
    for g in global_batches(n, B, world):
        for i in rank_slice(g, B, rank):
            hyp[i] = (i, 2 * i, 7)
    hyp = sync.allreduce_numpy(hyp)
    ok = bool((hyp == np.stack([np.arange(n), 2 * np.arange(n), np.full(n, 7)], 1)).all())
    q.put((steps, float(max(errs)), ok))
    dist.destroy_process_group()


def test_uneven_shards_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for steps, err, ok in res:
        assert steps == 3                  # ceil(9 / (2 * 2)) on BOTH ranks
        assert err < 1e-6                  # sum of globally-normalised shard gradients == gradient of the global batch
        assert ok


def test_unique_id_bootstrap_through_the_launchers_store(tmp_path):
    """bench.py / fit() under `python -m torch.distributed.run`: rank 0's RCCL unique id reaches the other ranks through the
    store the launcher already hosts on MASTER_PORT (no second port), and through a rank-0-hosted TCPStore on
    MASTER_PORT + 1 when the processes were started by hand."""
    import subprocess, sys, socket
    script = tmp_path / 'boot.py'
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ecog2txt_amd import parallel\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "import time\n"
        "got = parallel.share_from_rank0(bytes(range(128)) if rank == 0 else None, rank, world)\n"
        "assert got == bytes(range(128)), got\n"
        "# a SECOND bootstrap in the same job (a new communicator after the subject set changed): rank 0 is late (it is still\n"
        "# writing a checkpoint, say) -- the early rank must wait for the new id, not read the first one again\n"
        "if rank == 0: time.sleep(1.5)\n"
        "got2 = parallel.share_from_rank0(bytes(range(1, 129)) if rank == 0 else None, rank, world)\n"
        "assert got2 == bytes(range(1, 129)), got2\n"
        "print('rank', rank, 'ok', flush=True)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    def free_port():
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            return s.getsockname()[1]
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(free_port()), str(script)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout.count('ok') == 2, out.stdout + out.stderr
    port = free_port()
    env = dict(os.environ, WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.pop('TORCHELASTIC_USE_AGENT_STORE', None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in (1, 0)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all('ok' in o for o in outs), outs
