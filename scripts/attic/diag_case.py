import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import seq2seq as O
from test_gpu_parity import build, SPECS
name = sys.argv[1]; B, T, L = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = SPECS[name]
eng, ws, ospec, P, batch = build(kw, B, T, L, seed=4, ragged=False)
train = kw['ff_dropout'] > 0 or kw['rnn_dropout'] > 0
eng.forward(ws, train=train); eng.backward(ws, train=train); torch.cuda.synchronize()
want, cache = O.forward(P, ospec, batch, train=train, seed=11, emulate_bf16=True)
print(eng.losses(ws), want)
G = O.backward(P, cache); Gd = eng.store.export_tf('g')
for k in sorted(G):
    s = np.abs(G[k]).max() + 1e-12
    e = np.abs(Gd[k] - G[k])
    print('%-60s err/max %.2e  max %.3e  argmax %s' % (k, e.max() / s, s, np.unravel_index(e.argmax(), e.shape)))
k = [k for k in G if 'projection' in k and k.endswith('_0/biases') and 'encoder' in k]
if k:
    print(np.round(Gd[k[0]][:12], 6), np.round(G[k[0]][:12], 6))
