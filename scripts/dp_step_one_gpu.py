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
batch = bench.synth_batch(kw, B, T, L, seed=5)
for mode in (os.environ.get('DP_MODES', 'one_graph,graph_per_stage,single').split(',')):
    # (one engine per schedule: both data-parallel schedules -- the step as ONE graph with the collectives as nodes, and the default,
    #  one graph per backward stage with the collectives issued between them -- and the single-GPU graph for reference)
    eng = Seq2SeqEngine(NetSpec(**kw), device='cuda:0', seed=3, options={'dp_one_graph': mode == 'one_graph'})
    eng.init_params(seed=0)
    ws = eng.workspace(list(kw['channels'])[0], B, T, L)
    eng.set_batch(ws, batch)
    ntok, nval = eng.local_counts(batch['decoder_targets'], batch['encoder_targets'])
    eng.set_global_counts(ws, ntok, nval)
    sync = None
    if mode != 'single':
        sync = RcclSync(eng.store.g, 0, 1, RcclSync.unique_id(), 0, sum_of_global_means=True)
        sync.world = 2
    with eng.on_step_stream():
        for _ in range(5):
            eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(ws, use_graph=True, sync=sync)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    label = {'one_graph': 'data-parallel schedule, ONE graph with the collectives as nodes (one rank as two)',
             'graph_per_stage': 'data-parallel schedule, one graph per backward stage + eager collectives (the default; one rank as two)',
             'single': 'single-GPU graph'}[mode]
    print('%s %s: %.3f ms per step' % (cfg, label, 1e3 * dt), flush=True)
    if sync is not None:
        sync.close()
    eng._ws.clear()
    del eng, ws
    torch.cuda.empty_cache()
