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
