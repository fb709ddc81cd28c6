"""Diagnostic: where does the persistent recurrence differ from the launch-per-step path?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_gpu_parity import build, SPECS
name, B, T, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
outs = []
for flag in ('0', '1'):
    os.environ['E2T_PERSISTENT'] = flag
    eng, ws, ospec, P, batch = build(SPECS[name], B, T, L, seed=4, ragged=True)
    eng.forward(ws, train=True); torch.cuda.synchronize()
    outs.append([lw['Yext'].float().cpu().numpy() for lw in ws['enc']] + [ws['lens_d'].cpu().numpy(), int(eng.sync_err[0].item())])
lens = outs[0][-2]
print('err', outs[1][-1], 'lens', lens[:16], 'S', ws['S'])
for l in range(len(eng.enc)):
    a, b = outs[0][l], outs[1][l]
    Bq = B; S = ws['S']
    a = a.reshape(S + 3, Bq, -1); b = b.reshape(S + 3, Bq, -1)
    bad = np.argwhere(a != b)
    print('layer', l, 'H', eng.enc[l].H, 'mismatches', len(bad), 'of', a.size)
    if len(bad):
        print('  first', bad[:8].tolist())
        print('  blocks', np.unique(bad[:, 0]), 'rows', np.unique(bad[:, 1])[:20], 'cols', np.unique(bad[:, 2])[:40])
        i = tuple(bad[0]); print('  values step/persist', a[i], b[i], 'maxabs', np.abs(a - b).max())
    break
