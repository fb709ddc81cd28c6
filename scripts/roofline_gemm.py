"""Launch only the dominant GEMM instance of the cfg2 train step (encoder input projection, Gx = Ydrop . Wx^T) so
that a rocprofv3 --pmc / --kernel-trace run over this script gives per-launch numbers for exactly that instance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
kw, B, T, L = bench.CONFIGS['cfg2']
eng = Seq2SeqEngine(NetSpec(**kw), seed=1)
eng.init_params(0)
eng.pack('p')
ws = eng.workspace(401, B, T, L)
eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
eng.forward(ws, train=True)
torch.cuda.synchronize()
lay, lw = eng.enc[1], ws['enc'][1]
x = ws['enc'][0]['Ydrop'].data_ptr()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    eng.gemm(x, lay.in_ld, lay.WxT.data_ptr(), lay.in_ld, lw['Gx'].data_ptr(), lay.N4, ws['M'], lay.N4, lay.in_ld, bias=lay.bias_ptr(eng.store.p))
torch.cuda.synchronize()
print('M N K', ws['M'], lay.N4, lay.D + 1)
