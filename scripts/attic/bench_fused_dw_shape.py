"""Would ONE K-major GEMM per direction over [x | 1 | h_prev] (M = 1201) beat the separate dW_x (801 x 3200) and batched
dW_h (2 x 400 x 1600) products?  Measured: 134.6 vs 137.3 us -- no (tile waste moves, the K loop dominates)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
def r8(x): return (x + 7) // 8 * 8
wsb = torch.zeros(32 * 1024 * 1024, device='cuda')
def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
K = 8704
def tn(M, N, batch):
    lda, ldb = (M + 63) // 64 * 64 * batch, (N + 63) // 64 * 64 * batch
    a = torch.randn(K, lda, device='cuda').to(torch.bfloat16); b = torch.randn(K, ldb, device='cuda').to(torch.bfloat16)
    c = torch.zeros(batch * M * r8(N), device='cuda')
    ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = H.GEMM_SPLITK
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    if batch > 1:
        ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = batch, lda // batch, ldb // batch, M * r8(N)
    return timeit(lambda: lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), r8(N), M, N, K, C.byref(ep), torch.cuda.current_stream().cuda_stream))
t1 = tn(801, 3200, 1); t2 = tn(400, 1600, 2); t3 = tn(1201, 1600, 2)
print('dWx %.1f us + dWh(batched) %.1f us = %.1f us   vs fused [x|1|h]^T dG per direction, batched: %.1f us' % (t1, t2, t1 + t2, t3))
