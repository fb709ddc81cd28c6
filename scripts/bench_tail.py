"""The early optimiser update + re-pack of a configuration's step ALONE (no other kernel on the chip): fused (e2t_adam_pack_batch)
against separate (e2t_adam_ema_step + e2t_pack_batch), over the engine's real descriptor tables.  usage: bench_tail.py [cfg2|cfg4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, capture

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
eng.pack('p')
eng.store.g.normal_()
eng.store.v.uniform_(0.0, 1e-3)
hi = eng.store.seg_range('enc0.Wx')[0]
er = [(a, min(b, hi)) for a, b in eng.trainable_ranges(401) if a < hi]
npar = sum(b - a for a, b in er)
eng._pack_subtable(tuple(er)); plan = eng._fused_update_plan(tuple(er))
print('%s: early ranges %.1f M parameters; tile descriptors %s (+%s pack-only), %d workgroups; plain ranges %s' % (
    cfg, npar / 1e6, plan[0][1] if plan[0] else 0, plan[1][1] if plan[1] else 0, plan[0][2] if plan[0] else 0, plan[2][:6]))


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with capture(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


with eng.on_step_stream():
    t_f = timeit(lambda: eng.adam_pack_ranges(er, step_offset=1))
    t_a = timeit(lambda: eng.adam_ranges(er, step_offset=1))
    t_p = timeit(lambda: eng.pack_ranges(er))
print('  fused %.1f us (%.2f TB/s over 40 B/param); separate: update %.1f + re-pack %.1f = %.1f us (%.2f TB/s over 48 B/param)' % (
    t_f, 40 * npar / t_f / 1e6, t_a, t_p, t_a + t_p, 48 * npar / (t_a + t_p) / 1e6))
