"""Per-tensor HIP-vs-oracle errors of the cfg2 graph at B = 16, dropout off / on, plus the oracle's own bf16-vs-fp64 band
(diagnostic; run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from oracle import seq2seq as O
from test_gpu_fullsize_parity import _hip, _ragged, _biases_off_zero
from ecog2txt_amd.engine import NetSpec
kw, _, T, L = bench.CONFIGS['cfg2']
B = 16
ospec = O.NetSpec(**NetSpec(**kw).as_dict())
P = _biases_off_zero(O.init_params(ospec, seed=5), 6)
batch = bench.synth_batch(kw, B, T, L, seed=9)
_ragged(batch, T, 200, seed=2)
for train in (False, True):
    eng, ws, losses, logits, G = _hip(kw, B, T, L, batch, P, train=train)
    want, cache = O.forward(P, ospec, batch, train=train, seed=5, emulate_bf16=True)
    WG = O.backward(P, cache)
    ex, cache_x = O.forward(P, ospec, batch, train=train, seed=5, emulate_bf16=False)
    XG = O.backward(P, cache_x)
    print('train', train, 'losses', losses, {k: round(v, 6) for k, v in want.items()})
    for k in sorted(WG):
        s = np.abs(WG[k]).max() + 1e-12
        e = np.abs(G[k] - WG[k]) / s
        ex_ = np.abs(XG[k] - WG[k]) / s
        print('  %-64s max %.2e  >5e-3: %.4f  relL2 %.2e | emu-vs-fp64 max %.2e >5e-3 %.4f relL2 %.2e' % (
            k, e.max(), (e > 5e-3).mean(), np.linalg.norm(G[k] - WG[k]) / (np.linalg.norm(WG[k]) + 1e-12), ex_.max(), (ex_ > 5e-3).mean(),
            np.linalg.norm(XG[k] - WG[k]) / (np.linalg.norm(WG[k]) + 1e-12)))
