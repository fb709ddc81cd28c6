"""Diagnostic: wide (decoder) persistent kernels at several batch sizes: cluster counts that are not a multiple of the 8
XCDs, partly filled row blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, _, T, L = bench.CONFIGS['cfg2']
for B in [int(x) for x in (sys.argv[1:] or ['256', '224', '200', '96', '40'])]:
    eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
    eng.init_params(0)
    ws = eng.workspace(401, B, 96, L)
    eng.set_batch(ws, bench.synth_batch(kw, B, 96, L, 1))
    res = []
    for it in range(4):
        eng.forward(ws, train=True); eng.backward(ws, train=True)
        torch.cuda.synchronize()
        res.append(eng.sync_err.cpu().numpy()[:8].tolist())
        eng.sync_err.zero_()
    print('B=%3d: dec wide fwd ok=%s bwd ok=%s  err per iteration: %s' % (B, eng.dec.persistent_ok(B, eng.num_cus), eng.dec.persistent_bwd_ok(B, eng.num_cus), res), flush=True)
