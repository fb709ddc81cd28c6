import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
for rep in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    first = None
    for i in range(20):
        eng.train_step(ws)
        if first is None and os.environ.get('PERSTEP'):
            e = int(eng.sync_err[0].item())
            l0 = float(ws['loss'][0].item())
            if e or l0 != l0: first = (rep * 20 + i + 1, e if e else -1)
    if first: print('   first error at step %d, code %d (1 fwd, 2 fwd wide, 3 bwd, -1 NaN loss without a timeout)' % first)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    err = int(eng.sync_err[0].item())
    l = ws['loss'].cpu().numpy()
    print('steps %3d: %.3f ms/step  err=%d  loss=%s  step_t=%d' % ((rep + 1) * 20, dt, err, l[:3], int(eng.step_t.item())), flush=True)
