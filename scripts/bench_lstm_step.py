"""Micro-benchmark: forward/backward recurrence of one encoder layer (S graph-captured step launches)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, ceil_div
from ecog2txt_amd.hip_lib import lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
kw, B, T, L = bench.CONFIGS[cfg]
spec = NetSpec(**kw)
eng = Seq2SeqEngine(spec, seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True); eng.backward(ws, train=True)
torch.cuda.synchronize()
S = ws['S']
def timeit(fn, n, reps=20):
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
    return e0.elapsed_time(e1) * 1e3 / (reps * n)
lay, lw = eng.enc[1], ws['enc'][1]
d = lay.desc(lw, True)
def fwd():
    lib.e2t_lstm_seq_fwd(C.byref(d), lw['Gx'].data_ptr(), lay.WhF.data_ptr(), lw['Yext'].data_ptr(), lw['Ydrop'].data_ptr(),
                         lw['Cs'].data_ptr(), lw['Gs'].data_ptr(), ws['lens_d'].data_ptr(), None, 0, S, eng.stream)
def bwd():
    lib.e2t_lstm_seq_bwd(C.byref(d), lay.WhB.data_ptr(), lw['dG'].data_ptr(), lw['dG'].shape[1], ws['dY'][1].data_ptr(), lay.ldy,
                         lw['Gs'].data_ptr(), lw['Cs'].data_ptr(), ws['lens_d'].data_ptr(), None, None, None,
                         lw['dc_carry'].data_ptr(), None, None, eng.stream)
cnt = torch.zeros(4096, dtype=torch.int32, device='cuda'); err = torch.zeros(1, dtype=torch.int32, device='cuda')
def fwd_p():
    lib.e2t_lstm_seq_fwd_persistent(C.byref(d), lw['Gx'].data_ptr(), lay.WhF.data_ptr(), lw['Yext'].data_ptr(), lw['Ydrop'].data_ptr(),
                                    lw['Cs'].data_ptr(), lw['Gs'].data_ptr(), ws['lens_d'].data_ptr(), None, lw['hx'].data_ptr(),
                                    err.data_ptr(), eng.num_cus, eng.stream)
if lay.persistent_ok(B, eng.num_cus):
    print('persistent fwd: %.2f us/step (S=%d), err=%d' % (timeit(fwd_p, S), S, int(err.item())), flush=True)
    if os.environ.get('TIMELINE'):
        import numpy as np
        dbg = torch.zeros(256 * 8 * 8 + 2048, dtype=torch.int64, device='cuda')
        os.environ['E2T_LSTM_DBG'] = str(dbg.data_ptr())
        fwd_p(); torch.cuda.synchronize()
        del os.environ['E2T_LSTM_DBG']
        raw = dbg.cpu().numpy()
        nw = 4 * ceil_div(B, 64) * lay.ndir * lay.UT
        tot = raw[nw * 8: nw * 8 + nw] / 100.0
        first = raw[:nw * 8].reshape(-1, 8)[:, 7] / 100.0
        print('  whole kernel per wave: min %.1f med %.1f max %.1f us; prologue+step0: med %.1f max %.1f us' % (tot.min(), np.median(tot), tot.max(), np.median(first), first.max()))
        t = raw[:nw * 8].reshape(-1, 8)[:, :7]
        t = t[t[:, 0] > 0]
        spins = t[:, 1].copy()
        print('  retries of the state loads at that step: mean %.2f, share of waves with >= 1: %.2f, max %d' % (spins.mean(), (spins > 0).mean(), spins.max()))
        t[:, 1] = t[:, 0]
        rel = (t - t[:, :1]) / 100.0
        names = ['step top', '(unused)', 'state landed (incl. retries)', 'mma done', 'h stored', '(unused)', 'side work issued']
        print('  persistent step %d, %d waves; time since step top (us):' % (S // 2, len(t)))
        for i, nme in enumerate(names):
            print('    %-20s min %.2f  median %.2f  max %.2f' % (nme, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
        dd = np.diff(rel, axis=1)
        print('    phase durations: ' + ' | '.join('%s: min %.1f med %.1f p90 %.1f max %.1f' % (names[i + 1], dd[:, i].min(), np.median(dd[:, i]), np.percentile(dd[:, i], 90), dd[:, i].max()) for i in range(6)))
        print('    spread of step-top across waves: %.2f us (100 MHz wall clock: values are us)' % ((t[:, 0].max() - t[:, 0].min()) / 100.0))
def bwd_p():
    lib.e2t_lstm_seq_bwd_persistent(C.byref(d), lay.WhB.data_ptr(), lw['dG'].data_ptr(), lw['dG'].shape[1], ws['dY'][1].data_ptr(), lay.ldy,
                                    lw['Gs'].data_ptr(), lw['Cs'].data_ptr(), ws['lens_d'].data_ptr(), None, None, None, None, None,
                                    lw['dgx'].data_ptr(), cnt.data_ptr(), err.data_ptr(), eng.num_cus, eng.stream)
if lay.persistent_bwd_ok(B, eng.num_cus):
    print('persistent bwd: %.2f us/step (S=%d), err=%d' % (timeit(bwd_p, S), S, int(err.item())), flush=True)
    if os.environ.get('TIMELINE'):
        import numpy as np
        dbg = torch.zeros(256 * 8 * 8 + 2048, dtype=torch.int64, device='cuda')
        os.environ['E2T_LSTM_DBG'] = str(dbg.data_ptr())
        bwd_p(); torch.cuda.synchronize()
        del os.environ['E2T_LSTM_DBG']
        nw = 4 * ceil_div(B, 16) * lay.ndir * ceil_div(lay.UT, 4)
        t = dbg.cpu().numpy()[:nw * 8].reshape(-1, 8)[:, :7]
        t = t[t[:, 0] > 0]
        rel = (t - t[:, :1]) / 100.0
        names = ['step top', 'state landed (incl. retries)', '(same)', 'mma+reduce done', 'dG exchange stored', '(same)', 'side work done']
        dd = np.diff(rel, axis=1)
        print('  persistent bwd step %d, %d waves; phase durations (us): ' % (S // 2, len(t)) + ' | '.join('%s: min %.1f med %.1f p90 %.1f max %.1f' % (names[i + 1], dd[:, i].min(), np.median(dd[:, i]), np.percentile(dd[:, i], 90), dd[:, i].max()) for i in range(6)))
print('launch per step: fwd %.2f us/step   bwd %.2f us/step' % (timeit(fwd, S), timeit(bwd, S)), flush=True)
# ---- per-phase timeline of ONE step (s_memtime stamps written by the kernel) ----
if os.environ.get('TIMELINE'):
    import numpy as np
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
    os.environ['E2T_LSTM_DBG'] = str(dbg.data_ptr())
    lib.e2t_lstm_seq_fwd(C.byref(d), lw['Gx'].data_ptr(), lay.WhF.data_ptr(), lw['Yext'].data_ptr(), lw['Ydrop'].data_ptr(),
                         lw['Cs'].data_ptr(), lw['Gs'].data_ptr(), ws['lens_d'].data_ptr(), None, 5, 6, eng.stream)
    torch.cuda.synchronize()
    del os.environ['E2T_LSTM_DBG']
    t = dbg.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    rel = (t - base) / 100.0            # s_memtime ticks at 100 MHz -> us
    names = ['entry', 'lens ready', 'bulk issued', 'dma landed', 'barrier passed', 'mma done', 'math done', 'stores issued']
    print('waves recorded', len(t))
    for i, nme in enumerate(names):
        print('  %-15s  min %.2f  median %.2f  max %.2f us' % (nme, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
    dd = np.diff(rel, axis=1)
    print('  per-wave phase durations (median):', ' | '.join('%s %.2f' % (names[i + 1], np.median(dd[:, i])) for i in range(7)))
