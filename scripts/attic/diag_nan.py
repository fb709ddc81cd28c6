"""Diagnostic: run train steps and report the first tensor that stops being finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
graph = os.environ.get('GRAPH', '1') == '1'
def bad(t):
    return not bool(torch.isfinite(t.float()).all())
for step in range(1, 201):
    eng.train_step(ws, use_graph=graph)
    torch.cuda.synchronize()
    e = int(eng.sync_err[0].item())
    checks = [('loss', ws['loss'])]
    for l, lw in enumerate(ws['enc']):
        checks += [('enc%d.Gx' % l, lw['Gx']), ('enc%d.Yext' % l, lw['Yext']), ('enc%d.dG' % l, lw['dG'])]
    checks += [('dec.Gx', ws['dec']['Gx']), ('dec.Yext', ws['dec']['Yext']), ('dec.dG', ws['dec']['dG']), ('logits', ws['proj']['out']),
               ('aux.out', ws['aux']['out']), ('dP', ws['dP']), ('dlogits', ws['dlogits']), ('dHd', ws['dHd']), ('dh0', ws['dh0']), ('g', eng.store.g), ('p', eng.store.p)]
    bads = [n for n, t in checks if bad(t)]
    if e or bads:
        print('step %d: err=%d, non-finite: %s' % (step, e, bads))
        if 'g' in bads:
            g = eng.store.g
            for nm in eng.store.order:
                a, b = eng.store.seg_range(nm)
                if bad(g[a:b]): print('   grad segment', nm)
        break
else:
    print('200 steps clean')
