"""Soak test: thousands of train steps with ragged, changing batches, interleaved evaluation (forward + greedy decode with
the EMA weights), for cfg2 and a long-sequence variant; any in-kernel timeout, slow step or non-finite loss is reported."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
secs = float(os.environ.get('SOAK_SECONDS', '20'))
for cfg, B, T in (('cfg2', 256, 400), ('cfg2', 200, 1212), ('cfg2', 64, 96), ('cfg5', 256, 2000), ('cfg4', 256, 400)):
    kw, _, _, L = bench.CONFIGS[cfg]
    eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
    eng.init_params(0)
    ws = eng.workspace(list(kw['channels'])[0], B, T, L)
    batch = bench.synth_batch(kw, B, T, L, 1)
    eng.set_batch(ws, batch)
    X0 = ws['X'].clone()
    rng = np.random.default_rng(0)
    t_end = time.time() + secs
    steps = slow = 0
    while time.time() < t_end:
        if steps % 7 == 0:            # new ragged lengths (incl. empty and full utterances)
            lens = rng.integers(0, T + 1, size=B); lens[rng.integers(0, B)] = T; lens[rng.integers(0, B)] = 0
            mask = (torch.arange(T, device='cuda')[None, :] < torch.tensor(lens, device='cuda')[:, None]).float()[:, :, None]
            ws['X'].copy_(X0 * mask + (X0 == 0).float() * mask * 1e-3)       # keep valid rows non-zero
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.train_step(ws)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps += 1
        if steps > 3 and dt > 50e-3: slow += 1
        if steps % 50 == 0:
            eng.forward(ws, train=False, which='ema'); hyp = eng.greedy_decode(ws, which='ema'); torch.cuda.synchronize()
        e = eng.sync_err.cpu().numpy()
        if e[0]:
            print('%s B=%d T=%d: TIMEOUT at step %d: %s' % (cfg, B, T, steps, e[:8].tolist())); break
    l = eng.losses(ws)
    print('%s B=%d T=%d (S=%d): %d steps, %d slow, final loss %.4f, finite %s' % (cfg, B, T, ws['S'], steps, slow, l['total'], bool(np.isfinite(l['total']))), flush=True)
