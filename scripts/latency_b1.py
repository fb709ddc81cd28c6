"""Single-utterance decoding latency (the reference's online predictor, ecog2txt/trainers.py:925-949): one
zero-padded utterance of raw ECoG on the HOST -> greedy word ids on the host, with the EMA weights, exactly what
SequenceNetwork.online_predictor's predict() does per call (host->device copy, encoder + L decoder steps, device->host copy,
error-word check).  Eager launches and the decode replayed from one captured graph; B = 1 and, for scale, B = 8.
usage: latency_b1.py [cfg2|cfg4] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kw, _, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
eng.pack('ema')
sweep = os.environ.get('LAT_SWEEP') == '1'            # B = 1, 2, 4, 8 with the one-launch head (e2t_greedy_head_small) on and off, eager only
for B in ((1, 2, 4, 8) if sweep else (1, 8)):
    ws = eng.workspace(401, B, T, L)
    x = bench.synth_batch(kw, B, T, L, seed=3)['encoder_inputs'].astype(np.float32)
    x[:, 350:] = 0                                      # 1.75 s of signal, zero padded to the 2-s window
    xh = torch.from_numpy(x).pin_memory()
    for mode in (('eager', 'eager/general-head') if sweep else ('eager', 'graph')):
        eng.options['small_batch_head'] = mode != 'eager/general-head'
        def predict():
            ws['X'].copy_(xh, non_blocking=True)
            hyp = eng.greedy_decode(ws, which='ema', use_graph=(mode == 'graph')).cpu().numpy()
            eng.check_sync(ws)
            return hyp
        for _ in range(10):
            ref = predict()
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter(); h = predict(); ts.append(time.perf_counter() - t0)
            assert np.array_equal(h, ref)
        ts = 1e3 * np.sort(np.array(ts))
        print('%s B=%d %-18s  host-to-host per call: median %.3f ms  p10 %.3f  p99 %.3f   (%.0f utterances/s)' % (
            cfg, B, mode, np.median(ts), ts[len(ts) // 10], ts[int(0.99 * len(ts))], B / np.median(ts) * 1e3), flush=True)
