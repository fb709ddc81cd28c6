"""Micro-benchmark: decoder recurrences (H=800, L steps) -- persistent wide kernels vs launch-per-step."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
from ecog2txt_amd.hip_lib import lib
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True); eng.backward(ws, train=True)
torch.cuda.synchronize()
lay, lw = eng.dec, ws['dec']
S = lw['S']
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
x = ws['e'].data_ptr()
for mode in ('1', '0'):
    eng.persistent_fwd = eng.persistent_bwd = mode == '1'
    f = timeit(lambda: lay.fwd(lw, x, ws['dlens'], eng.store.p, True, c0=ws['c0'], steps=(0, S)))
    b = timeit(lambda: lay.bwd_rec(lw, x, ws['dlens'], ws['dHd'].data_ptr(), lay.ldy, True, None, 0, c0=ws['c0'], dh0=ws['dh0'], dc0=ws['dc0']))
    print('decoder S=%d persistent=%s: fwd %.1f us (%.2f us/step)   bwd %.1f us (%.2f us/step incl. pseudo-step)' % (S, mode, f, f / S, b, b / (S + 1)))
assert int(eng.sync_err[0].item()) == 0
