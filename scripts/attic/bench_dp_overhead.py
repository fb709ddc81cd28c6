"""Cost of the data-parallel step structure on ONE GPU: staged graphs + side graphs with a no-op gradient exchange vs the
single captured graph (what remains for a real run is the all-reduce itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
class NoSync:
    world, grad_scale = 2, 1.0
    def allreduce_range(self, a, b): return None
    def wait(self): pass
kw, B, T, L = bench.CONFIGS['cfg2']
for name, sync in (('single graph', None), ('staged (data-parallel) graphs, no-op exchange', NoSync()), ('single graph', None),
                   ('staged (data-parallel) graphs, no-op exchange', NoSync())):
    eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
    eng.init_params(0)
    ws = eng.workspace(401, B, T, L)
    eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
    for _ in range(5): eng.train_step(ws, sync=sync)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): eng.train_step(ws, sync=sync)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    print('%-50s %s ms/step' % (name, ' '.join('%.3f' % t for t in ts)))
