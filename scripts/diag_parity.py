"""Print actual HIP-vs-oracle errors (diagnostic; run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import seq2seq as O
from test_gpu_parity import build, SPECS
for name in ['small_dropout', 'mid']:
    B, T, L = (40, 100, 8) if name == 'mid' else (19, 26, 6)
    eng, ws, ospec, P, batch = build(SPECS[name], B, T, L, seed=4, ragged=True)
    eng.forward(ws, train=True); eng.backward(ws, train=True); torch.cuda.synchronize()
    got = eng.losses(ws)
    want, cache = O.forward(P, ospec, batch, train=True, seed=11, emulate_bf16=True)
    exact, cache_x = O.forward(P, ospec, batch, train=True, seed=11, emulate_bf16=False)
    print(name, 'gpu', got); print(name, 'emu', {k: round(v, 6) for k, v in want.items()}); print(name, 'fp64', {k: round(v, 6) for k, v in exact.items()})
    G = O.backward(P, cache); Gx = O.backward(P, cache_x); Gd = eng.store.export_tf('g')
    for k in sorted(G):
        s = np.abs(G[k]).max() + 1e-12
        print('  %-62s gpu-vs-emu %.2e   emu-vs-fp64 %.2e' % (k, np.abs(Gd[k] - G[k]).max() / s, np.abs(Gx[k] - G[k]).max() / s))
import __graft_entry__ as g
g.smoke()
