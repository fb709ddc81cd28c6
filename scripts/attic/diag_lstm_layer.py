"""Diagnostic: one bi-LSTM layer forward on the GPU vs the oracle; prints where they differ."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import seq2seq as O
from oracle.bf16 import round_bf16, from_bf16_bits
from test_gpu_parity import build, SPECS
name = sys.argv[1] if len(sys.argv) > 1 else 'small_dropout'
eng, ws, ospec, P, batch = build(SPECS[name], 19, 26, 6, seed=4, ragged=len(sys.argv) > 2)
eng.forward(ws, train=False); torch.cuda.synchronize()
want, cache = O.forward(P, ospec, batch, train=False, emulate_bf16=True)
S, B = ws['S'], ws['B']
for l, lay in enumerate(eng.enc):
    lw = ws['enc'][l]
    H, H8 = lay.H, lay.H8
    Y = lw['Yext'].view(torch.int16).cpu().numpy().view(np.uint16)
    Y = from_bf16_bits(Y).reshape(S + 3, B, lay.ldy)[1:S + 1]
    for d, dn in enumerate(('fw', 'bw')):
        got = Y[:, :, d * H8:d * H8 + H]
        ref = cache['enc'][l][dn]['Yq']
        err = np.abs(got - ref)
        bad = np.argwhere(err > 1e-2)
        print('layer %d %s: max err %.3e, bad %d of %d' % (l, dn, err.max(), len(bad), err.size))
        if len(bad):
            print('   first bad (t,b,u):', bad[:12].tolist())
            print('   bad t set', sorted(set(bad[:, 0].tolist()))[:20], ' bad u set', sorted(set(bad[:, 2].tolist()))[:20], ' bad b set', sorted(set(bad[:, 1].tolist()))[:20])
    E = from_bf16_bits(ws['E'].view(torch.int16).cpu().numpy().view(np.uint16)).reshape(S, B, -1)[:, :, :ospec.enc_embed]
    if l == 0:
        print('conv out max err %.3e' % np.abs(E - cache['E']).max())
print('lens', ws['lens_d'].cpu().numpy().tolist())
print('losses', eng.losses(ws), want)
