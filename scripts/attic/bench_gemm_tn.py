"""Weight-gradient shapes: TN GEMM on the K-major operands vs NT GEMM on pre-transposed copies (+ the transposes)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
def r8(x): return (x + 7) // 8 * 8
wsb = torch.zeros(16 * 1024 * 1024, device='cuda')
def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
for name, M, N, K in [('dWx enc l1/l2', 801, 3200, 8704), ('dWh one dir', 400, 1600, 8704), ('dWx enc l0', 101, 3200, 8704), ('conv dW', 3073, 100, 8704), ('dWproj', 801, 1806, 2560)]:
    lda, ldb = (M + 63) // 64 * 64, (N + 63) // 64 * 64
    a = torch.randn(K, lda, device='cuda').to(torch.bfloat16)
    b = torch.randn(K, ldb, device='cuda').to(torch.bfloat16)
    aT = a.t().contiguous(); bT = b.t().contiguous()
    c = torch.zeros(M, r8(N), device='cuda'); c2 = torch.zeros(M, r8(N), device='cuda')
    ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = H.GEMM_SPLITK
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    s = lambda: torch.cuda.current_stream().cuda_stream
    tn = lambda: lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), r8(N), M, N, K, C.byref(ep), s())
    nt = lambda: lib.e2t_gemm_nt_bf16(aT.data_ptr(), K, bT.data_ptr(), K, c2.data_ptr(), r8(N), M, N, K, C.byref(ep), s())
    def tr():
        lib.e2t_transpose_bf16(a.data_ptr(), lda, K, M, aT.data_ptr(), K, s()); lib.e2t_transpose_bf16(b.data_ptr(), ldb, K, N, bT.data_ptr(), K, s())
    t_tn, t_nt, t_tr = timeit(tn), timeit(nt), timeit(tr)
    err = (c[:, :N] - c2[:, :N]).abs().max().item() / c2.abs().max().item()
    print('%-14s M%5d N%5d K%5d  TN %7.1f us (%6.1f TF)   NT %7.1f us + transposes %6.1f us   max diff %.1e' % (name, M, N, K, t_tn, 2.0 * M * N * K / t_tn / 1e6, t_nt, t_tr, err), flush=True)
