"""The kernels that dominate the cfg2 train step, launched in isolation so that rocprofv3 --pmc passes over this script
give per-launch counters for the FINAL code: every GEMM product of one step (the three k_gemm_nt instances + the split-K
reductions, replayed from the engine's launch log) and the persistent recurrences (encoder layer 1 forward / BPTT, decoder
forward / BPTT).  `python scripts/roofline_kernels.py [cfg] [reps]`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw, B, T, L = bench.CONFIGS[cfg]
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
sid = list(kw['channels'])[0]
ws = eng.workspace(sid, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng._gemm_log = []
eng.forward(ws, train=True)
eng.backward(ws, train=True)
torch.cuda.synchronize()
log, eng._gemm_log = eng._gemm_log, None
for _ in range(reps):
    for r in log:
        eng.gemm_replay(r)
    torch.cuda.synchronize()
S = ws['S']
li = 1 if len(eng.enc) > 1 else 0
lay, lw = eng.enc[li], ws['enc'][li]
x = ws['enc'][li - 1]['Ydrop'].data_ptr() if li else ws['E'].data_ptr()
for _ in range(reps):
    lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, S))
    lay.bwd_rec(lw, x, ws['lens_d'], ws['dY'][li].data_ptr(), lay.ldy, True, None, 0, dy_masked=lay.out_drop(True) is not None)
    eng.dec.fwd(ws['dec'], ws['e'].data_ptr(), ws['dlens'], eng.store.p, True, c0=ws['c0'], gx_done=True)
    eng.dec.bwd_rec(ws['dec'], ws['e'].data_ptr(), ws['dlens'], ws['dHd'].data_ptr(), eng.dec.ldy, True, None, eng.E8, c0=ws['c0'],
                    dh0=ws['dh0'], dc0=ws['dc0'], dy_masked=True)
    torch.cuda.synchronize()
eng.check_sync()
inst = {}
for r in log:
    d = inst.setdefault(r['inst'], dict(n=0, flops=0, alg_bytes=0))
    d['n'] += 1; d['flops'] += r['flops']; d['alg_bytes'] += r['in_bytes'] + r['out_bytes']
print('PRODUCTS', {k: dict(v, flops_per_launch=v['flops'] // v['n'], alg_bytes_per_launch=v['alg_bytes'] // v['n']) for k, v in inst.items()})
