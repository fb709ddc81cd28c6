"""Phase timeline of the large-hidden-size persistent recurrences (csrc/lstm_big.hip) at cfg4 sizes."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
import numpy as np
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, ceil_div, capture
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True); eng.backward(ws, train=True)
torch.cuda.synchronize()
S = ws['S']
lay, lw = eng.enc[1], ws['enc'][1]
x = ws['enc'][0]['Ydrop'].data_ptr()
def timeit(fn, n, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)
fwd = lambda: lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, S))
bwd = lambda: lay.bwd_rec(lw, x, ws['lens_d'], ws['dY'][1].data_ptr(), lay.ldy, True, None, 0, dy_masked=True)
names = {'fwd': ['step top', 'poll passed', 'state landed', 'mma done', 'cells done', 'published', 'side work issued'],
         'bwd': ['step top', 'poll passed', 'stream+mma done', 'reduced', 'cells done', 'published', 'side work issued']}
for nm, fn in (('fwd', fwd), ('bwd', bwd)):
    print('%s: %.2f us/step (S=%d) err=%s' % (nm, timeit(fn, S), S, eng.sync_err.cpu().numpy()[:1]), flush=True)
    dbg = torch.zeros(256 * 4 * 8 + 64, dtype=torch.int64, device='cuda')
    os.environ['E2T_LSTM_DBG'] = str(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    del os.environ['E2T_LSTM_DBG']
    t = dbg.cpu().numpy()[:256 * 4 * 8].reshape(-1, 8)[:, :7]
    t = t[t[:, 0] > 0]
    rel = (t - t[:, :1]) / 100.0
    dd = np.diff(rel, axis=1)
    print('  step %d, %d waves; phase durations (us): ' % (S // 2, len(t)))
    for i in range(6):
        print('    -> %-18s min %.2f med %.2f p90 %.2f max %.2f' % (names[nm][i + 1], dd[:, i].min(), np.median(dd[:, i]), np.percentile(dd[:, i], 90), dd[:, i].max()))
    print('    total med %.2f; spread of step-top across waves %.2f us' % (np.median(rel[:, 6]), (t[:, 0].max() - t[:, 0].min()) / 100.0))
    if nm == 'fwd':
        full = dbg.cpu().numpy()[:256 * 4 * 8].reshape(256, 4, 8)[:, :, :7]
        side = (full[:, :, 6] - full[:, :, 5]) / 100.0
        poll = (full[:, :, 1] - full[:, :, 0]) / 100.0
        print('    side work by wave: ', ' '.join('%.2f' % np.median(side[:, w]) for w in range(4)), ' p90:', ' '.join('%.2f' % np.percentile(side[:, w], 90) for w in range(4)))
        print('    side work by cluster (block %% 8): ', ' '.join('%.2f/%.2f' % (np.median(side[c::8]), side[c::8].max()) for c in range(8)))
        print('    poll by cluster: ', ' '.join('%.2f/%.2f' % (np.median(poll[c::8]), poll[c::8].max()) for c in range(8)))
        # absolute times within cluster 0: publish stamps and poll-passed stamps relative to the cluster's earliest step top
        c0 = full[0::8]
        t0 = c0[:, :, 0].min()
        print('    cluster 0: step top %.2f..%.2f, published %.2f..%.2f, side done %.2f..%.2f (us since first step top)' % (
            (c0[:, :, 0].min() - t0) / 100, (c0[:, :, 0].max() - t0) / 100, (c0[:, :, 5].min() - t0) / 100, (c0[:, :, 5].max() - t0) / 100,
            (c0[:, :, 6].min() - t0) / 100, (c0[:, :, 6].max() - t0) / 100))
