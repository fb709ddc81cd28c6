"""Where does a step of the persistent recurrences go?  Isolated us-per-step of encoder layer 1's forward sweep and BPTT with parts
of the SIDE WORK switched off (diagnostics build, E2T_REC_VARIANT; the results of such a run are wrong -- timing only):
  forward  1 = no Philox mask (dropped copy = plain copy), 2 = no gate / cell saves, 4 = no dropped copy at all, 8 = no row-major h
  BPTT     1 = no Philox mask, 2 = no row-major dG stores, 4 = no factor precompute at all
    python scripts/probe_rec_sidework.py [cfg2]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import _dbg  # noqa: F401
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
from ecog2txt_amd.hip_lib import lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True); eng.backward(ws, train=True)
torch.cuda.synchronize()
S = ws['S']
lay, lw = eng.enc[1], ws['enc'][1]
d = lay.desc(lw, True)
err = torch.zeros(16, dtype=torch.int32, device='cuda'); cnt = torch.zeros(4096, dtype=torch.int32, device='cuda')


def timeit(fn, n, reps=30):
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


def fwd_p():
    lib.e2t_lstm_seq_fwd_persistent(C.byref(d), lw['Gx'].data_ptr(), lay.WhF.data_ptr(), lw['Yext'].data_ptr(), lw['Ydrop'].data_ptr(),
                                    lw['Cs'].data_ptr(), lw['Gs'].data_ptr(), ws['lens_d'].data_ptr(), None, lw['hx'].data_ptr(),
                                    err.data_ptr(), eng.num_cus, eng.stream)


def bwd_p():
    lib.e2t_lstm_seq_bwd_persistent(C.byref(d), lay.WhB.data_ptr(), lw['dG'].data_ptr(), lw['dG'].shape[1], ws['dY'][1].data_ptr(), lay.ldy,
                                    lw['Gs'].data_ptr(), lw['Cs'].data_ptr(), ws['lens_d'].data_ptr(), None, None, None, None, None,
                                    lw['dgx'].data_ptr(), cnt.data_ptr(), err.data_ptr(), eng.num_cus, eng.stream)


quick = len(sys.argv) > 2 and sys.argv[2] == 'quick'
for name, fn, variants in (('forward', fwd_p, (0,) if quick else (0, 1, 2, 4, 8, 3, 7, 15, 0)), ('BPTT', bwd_p, (0,) if quick else (0, 1, 2, 4, 6, 7, 0))):
    # round 6: E2T_FWD_DEFER / E2T_BWD_DEFER = 0 selects the round-5 order (mask / factors in FRONT of the next step's state loads), 1 the
    # product's (under them); the variant switches act on the round-5 order only
    modes = [('product order', {}), ('DEFER: side work UNDER the next state loads', dict(E2T_FWD_DEFER='1', E2T_BWD_DEFER='1')),
             ('PIPE: fragments consumed as they land', dict(E2T_FWD_PIPE='1', E2T_BWD_PIPE='1'))]
    if os.environ.get('PROBE_MODES'):
        modes = [m for m in modes if m[0].split(':')[0].split(' ')[0] in os.environ['PROBE_MODES'].split(',')]
    for rep in range(3):
        for label, env in modes:
            for k in ('E2T_FWD_DEFER', 'E2T_BWD_DEFER', 'E2T_FWD_PIPE', 'E2T_BWD_PIPE'):
                os.environ[k] = env.get(k, '0')
            os.environ['E2T_REC_VARIANT'] = '0'
            print('%s, %s: %.3f us per step (S = %d, err %d)' % (name, label, timeit(fn, S), S, int(err[0].item())), flush=True)
    for k in ('E2T_FWD_PIPE', 'E2T_BWD_PIPE'):
        os.environ[k] = '0'
    os.environ['E2T_FWD_DEFER'] = os.environ['E2T_BWD_DEFER'] = '0'
    for v in variants:
        os.environ['E2T_REC_VARIANT'] = str(v)
        print('%s (round-5 order) variant %2d: %.3f us per step (S = %d, err %d)' % (name, v, timeit(fn, S), S, int(err[0].item())), flush=True)
for k in ('E2T_REC_VARIANT', 'E2T_FWD_DEFER', 'E2T_BWD_DEFER', 'E2T_FWD_PIPE', 'E2T_BWD_PIPE'):
    os.environ.pop(k, None)
