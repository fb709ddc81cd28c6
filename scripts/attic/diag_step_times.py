"""Diagnostic: wall time and in-kernel-timeout flag of every train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
for step in range(1, int(os.environ.get('NSTEPS', '120')) + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.train_step(ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    ev = eng.sync_err.cpu().numpy(); e = int(ev[0])
    if dt > 4.0 or e:
        print('step %3d: %.2f ms  err=%d  loss=%.4f  detail(block,wave,k,flag idx,flag val,fbase,epoch)=%s' % (step, dt, e, float(ws['loss'][0].item()), ev[1:8].tolist()), flush=True)
    if e:
        eng.sync_err.zero_()
print('done')
