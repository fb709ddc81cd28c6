import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
for _ in range(3):
    eng.train_step(ws)
ts = []
for step in range(6000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.train_step(ws)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert int(eng.sync_err[0].item()) == 0
ts = np.array(ts) * 1e3
print('6000 steps: median %.3f ms, p99 %.3f, p99.9 %.3f, max %.3f; steps > 5 ms: %s' % (np.median(ts), np.percentile(ts, 99), np.percentile(ts, 99.9), ts.max(), [(int(i), round(float(ts[i]), 2)) for i in np.nonzero(ts > 5)[0][:20]]))
