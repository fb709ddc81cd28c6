"""Time the operand re-pack (k_pack_batch) and the optimiser step of the cfg2 model on their own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
spec_kw, B, T, L = bench.CONFIGS[os.environ.get('CFG', 'cfg2')]
spec = NetSpec(**spec_kw)
eng = Seq2SeqEngine(spec, device='cuda:0', seed=1)
eng.init_params(seed=0)
eng.pack('p')
sid = list(spec.channels)[0]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print('params %.1f M, pack table: %d descriptors, %d workgroups' % (eng.store.p.numel() / 1e6, eng._pack_table[0], eng._pack_table[1]))
print('pack      %7.1f us' % timeit(lambda: eng.pack('p')))
print('adam+ema  %7.1f us' % timeit(lambda: eng.adam_step(sid)))
if os.environ.get('DUMP'):
    import ctypes
    from ecog2txt_amd import hip_lib as H
    raw = bytes(eng._pack_dev.cpu().numpy())
    n = eng._pack_table[0]
    descs = (H.PackDesc * n).from_buffer_copy(raw)
    for d in descs:
        print('kind %d  s0 %6d s1 %6d  d0 %5d d1 %5d  ld %5d  off%%4 %d  first %d' % (d.kind, d.s0, d.s1, d.d0, d.d1, d.ld, d.src_off % 4, d.first_block))
