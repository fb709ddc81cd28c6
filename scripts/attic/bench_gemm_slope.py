import os, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
for name, M, N, K in [('256 K832', 8704, 3200, 832), ('256 K3200', 8704, 3200, 3200), ('128 K832 N800', 8704, 800, 832), ('128 K3200 N800', 8704, 800, 3200)]:
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16); b = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    c = torch.zeros(M, N, device='cuda')
    ep = H.GemmEpilogue(); ep.alpha = 1.0
    run = lambda: lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N, K, C.byref(ep), torch.cuda.current_stream().cuda_stream)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-16s %8.1f us  %7.1f TF' % (name, us, 2.0 * M * N * K / us / 1e6), flush=True)
