"""Latency of the online predictor path (one utterance, cfg2 sizes): host ECoG [T,C] -> greedy word ids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, _, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
eng.pack('ema')
for B in (1, 8, 64):
    ws = eng.workspace(401, B, T, L)
    x = np.abs(np.random.default_rng(0).standard_normal((B, T, 256))).astype(np.float32)
    def predict():
        ws['X'].copy_(torch.from_numpy(x)); ws['Y'].zero_()
        return eng.greedy_decode(ws, which='ema').cpu().numpy()
    for _ in range(3): predict()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): predict()
    dt = (time.perf_counter() - t0) / n
    print('B=%3d: %.2f ms per call (%.2f ms per utterance), host copy + encode + %d greedy steps, eager launches' % (B, dt * 1e3, dt * 1e3 / B, L))
