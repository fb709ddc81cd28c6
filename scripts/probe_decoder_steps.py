"""Decoder recurrences (persistent wide kernels, H = 800) over several target lengths: launch overhead vs per-step time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, capture
kw, B, T, _ = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        for _ in range(4):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * reps)


res = []
for L in (5, 10, 20, 40):
    ws = eng.workspace(401, B, T, L)
    eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
    eng.forward(ws, train=True); eng.backward(ws, train=True)
    torch.cuda.synchronize()
    lay, lw = eng.dec, ws['dec']
    x = ws['e'].data_ptr()
    with eng.on_step_stream():
        f = timeit(lambda: lay.fwd(lw, x, ws['dlens'], eng.store.p, True, c0=ws['c0'], steps=(0, L), gx_done=True))
        b = timeit(lambda: lay.bwd_rec(lw, x, ws['dlens'], ws['dHd'].data_ptr(), lay.ldy, True, None, 0, c0=ws['c0'], dh0=ws['dh0'], dc0=ws['dc0']))
    res.append((L, f, b))
    print('L=%2d  fwd %.1f us   bwd %.1f us' % (L, f, b), flush=True)
(l0, f0, b0), (l1, f1, b1) = res[1], res[3]
print('per step: fwd %.2f us, bwd %.2f us; launch + prologue: fwd %.1f us, bwd %.1f us' % ((f1 - f0) / (l1 - l0), (b1 - b0) / (l1 - l0),
      f0 - l0 * (f1 - f0) / (l1 - l0), b0 - l0 * (b1 - b0) / (l1 - l0)))
eng.check_sync()
