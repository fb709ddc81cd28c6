"""Forward/backward recurrence of one encoder layer with 1/2/4 concurrent row-block chains."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1); eng.init_params(0)
ws = eng.workspace(401, B, T, L); eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True); eng.backward(ws, train=True); torch.cuda.synchronize()
lay, lw = eng.enc[1], ws['enc'][1]
x = ws['enc'][0]['Ydrop'].data_ptr()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for ch in (1, 2, 4):
    eng.chains = ch
    f = timeit(lambda: lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, ws['S'])))
    b = timeit(lambda: lay.bwd(lw, x, ws['lens_d'], ws['dY'][1].data_ptr(), lay.ldy, True, None, 0))
    print('chains %d: fwd recurrence %.1f us (%.2f us/step)   bwd recurrence+grad GEMMs %.1f us' % (ch, f, f / ws['S'], b))
