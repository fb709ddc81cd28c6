"""How well does a GEMM run NEXT to a persistent recurrence?  (round 5: before building products that stream behind a recurrence's
progress.)  For one encoder layer of the configuration: the forward recurrence, the BPTT, the next layer's input projection (Gx),
the input gradient (dX) and the weight-gradient group -- each alone, then the recurrence with one or two of the products on side
branches of ONE captured graph (no data dependency between them here: the products are replays of the logged launches).
usage: probe_corun.py [cfg2|cfg4|cfg5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, capture

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng._gemm_log = []
eng.forward(ws, train=True); eng.backward(ws, train=True)
torch.cuda.synchronize()
log, eng._gemm_log = eng._gemm_log, None
S = ws['S']
li = 1
lay, lw = eng.enc[li], ws['enc'][li]
x = ws['enc'][li - 1]['Ydrop'].data_ptr()
H4 = 4 * lay.H * lay.ndir


def pick(pred, what):
    r = [r for r in log if pred(r)]
    assert r, what
    print('%-14s %s M=%d N=%d K=%d batch %d splits %d %s' % (what, r[0]['inst'], r[0]['M'], r[0]['N'], r[0]['K'], r[0]['batch'], r[0]['splits'], r[0].get('desc', '')))
    return r[0]


gx = pick(lambda r: not r['tn'] and r['N'] == H4 and r['M'] == ws['M'] and r['K'] >= 2 * lay.H, 'Gx (layer l+1)')
dx = pick(lambda r: not r['tn'] and r['K'] == H4 and r['M'] == ws['M'], 'dX (layer l)')
dw = max([r for r in log if r['inst'] == 'tn128g'], key=lambda r: r['flops'])
print('%-14s %s' % ('dW group', dw.get('desc', '')))

fwd = lambda: lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, S))
bwd = lambda: lay.bwd_rec(lw, x, ws['lens_d'], ws['dY'][li].data_ptr(), lay.ldy, True, None, 0, dy_masked=lay.out_drop(True) is not None)
side = [torch.cuda.Stream() for _ in range(2)]


NREP = 4


def graph_of(main, sides=()):
    main(); [f() for f in sides]; torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        cur = torch.cuda.current_stream()
        for _ in range(NREP):                                 # (several rounds per replay: the launch gap of a replay is not the subject)
            ev = torch.cuda.Event(); ev.record(cur)
            main()                                            # (the critical branch's first node is created first: it keeps the parent's queue)
            joins = []
            for st, f in zip(side, sides):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    f()
                    e = torch.cuda.Event(); e.record(st)
                joins.append(e)
            for e in joins:
                cur.wait_event(e)
    return g


def time_graph(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * NREP)


R = lambda rec: (lambda: eng.gemm_replay(rec))
with eng.on_step_stream():
    t = {}
    for name, main, sides in [('fwd', fwd, ()), ('BPTT', bwd, ()), ('Gx', R(gx), ()), ('dX', R(dx), ()), ('dW', R(dw), ()),
                              ('fwd || Gx', fwd, (R(gx),)), ('BPTT || dW', bwd, (R(dw),)), ('BPTT || dX', bwd, (R(dx),)),
                              ('BPTT || dW || dX', bwd, (R(dw), R(dx))), ('dX || dW', R(dx), (R(dw),)), ('Gx || Gx', R(gx), (R(gx),))]:
        t[name] = time_graph(graph_of(main, sides))
        parts = [p.strip() for p in name.split('||')]
        serial = sum(t[p] for p in parts) if len(parts) > 1 else None
        print('%-20s %7.1f us%s' % (name, t[name], '   (serial sum %.1f, longest alone %.1f)' % (serial, max(t[p] for p in parts)) if serial else ''), flush=True)
eng.check_sync()
