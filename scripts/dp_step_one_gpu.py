"""The data-parallel step schedule on ONE GPU: RcclSync on a one-rank communicator reported as two ranks (what
tests/test_gpu_dp_contention.py runs, without the occupying kernels).  usage: dp_step_one_gpu.py [cfg] [steps]
(for a kernel timeline: rocprofv3 --kernel-trace ... -- python scripts/dp_step_one_gpu.py, then scripts/step_timeline.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
from ecog2txt_amd.parallel import RcclSync
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), device='cuda:0', seed=3, options={'dp_one_graph': True})
eng.init_params(seed=0)
ws = eng.workspace(list(kw['channels'])[0], B, T, L)
batch = bench.synth_batch(kw, B, T, L, seed=5)
eng.set_batch(ws, batch)
ntok, nval = eng.local_counts(batch['decoder_targets'], batch['encoder_targets'])
eng.set_global_counts(ws, ntok, nval)
for dp in (True, False):
    sync = None
    if dp:
        sync = RcclSync(eng.store.g, 0, 1, RcclSync.unique_id(), 0, sum_of_global_means=True)
        sync.world = 2
    with eng.on_step_stream():
        for _ in range(5):
            eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%s %s: %.3f ms per step' % (cfg, 'data-parallel schedule (one rank as two)' if dp else 'single graph', 1e3 * dt), flush=True)
    if sync is not None:
        sync.close()
