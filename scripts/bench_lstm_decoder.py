"""Micro-benchmark: decoder recurrences (cfg2: H = 800; cfg4: H = 2048), L steps -- persistent wide kernels vs launch-per-step.
    python scripts/bench_lstm_decoder.py [cfg2|cfg4]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
from ecog2txt_amd.hip_lib import lib
CFG = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
kw, B, T, L = bench.CONFIGS[CFG]
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
for mode in (('1', '0') if lay.H <= 832 else ('0',)):
    eng.persistent_fwd = eng.persistent_bwd = mode == '1'
    f = timeit(lambda: lay.fwd(lw, x, ws['dlens'], eng.store.p, True, c0=ws['c0'], steps=(0, S)))
    b = timeit(lambda: lay.bwd_rec(lw, x, ws['dlens'], ws['dHd'].data_ptr(), lay.ldy, True, None, 0, c0=ws['c0'], dh0=ws['dh0'], dc0=ws['dc0']))
    print(CFG + ' decoder S=%d persistent=%s: fwd %.1f us (%.2f us/step)   bwd %.1f us (%.2f us/step incl. pseudo-step)' % (S, mode, f, f / S, b, b / (S + 1)))
assert int(eng.sync_err[0].item()) == 0
if os.environ.get('TIMELINE'):
    # per-wave phase stamps of the persistent BPTT (wide instance) at step S/2
    import numpy as np
    eng.persistent_fwd = eng.persistent_bwd = True
    dbg = torch.zeros(256 * 8 * 8 + 2048, dtype=torch.int64, device='cuda')
    os.environ['E2T_LSTM_DBG'] = str(dbg.data_ptr())
    lay.bwd_rec(lw, x, ws['dlens'], ws['dHd'].data_ptr(), lay.ldy, True, None, 0, c0=ws['c0'], dh0=ws['dh0'], dc0=ws['dc0'])
    torch.cuda.synchronize()
    del os.environ['E2T_LSTM_DBG']
    t = dbg.cpu().numpy()[:200 * 4 * 8].reshape(-1, 8)[:, :7]
    t = t[t[:, 0] > 0]
    dd = np.diff((t - t[:, :1]) / 100.0, axis=1)
    names = ['state of row tile 0 landed', '(same)', 'mma + reduce', 'exchange stored', '(same)', 'side work']
    print('decoder BPTT step %d, %d waves; phase durations (us): ' % (S // 2, len(t)) +
          ' | '.join('%s: med %.2f p90 %.2f' % (names[i], np.median(dd[:, i]), np.percentile(dd[:, i], 90)) for i in range(6)))
