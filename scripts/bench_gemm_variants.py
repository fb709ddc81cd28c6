"""Sensitivity of e2t_gemm_nt_bf16 on the encoder input-projection shape: K tail, output dtype, K and N size."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
def r8(x): return (x + 7) // 8 * 8
st = torch.cuda.current_stream().cuda_stream
BIAS = 1 << 20
for name, M, N, K, flags in [('K832 bias', 8704, 3200, 832, BIAS), ('K832 sparseA', 8704, 3200, 832, 1 << 21), ('K768', 8704, 3200, 768, 0), ('K800', 8704, 3200, 800, 0), ('K808', 8704, 3200, 808, 0), ('K832', 8704, 3200, 832, 0),
                             ('K808 bf16out', 8704, 3200, 808, H.GEMM_OUT_BF16), ('K104', 8704, 3200, 104, 0), ('K104 bf16out', 8704, 3200, 104, H.GEMM_OUT_BF16),
                             ('K64', 8704, 3200, 64, 0), ('K1600', 8704, 3200, 1600, 0), ('K3200', 8704, 3200, 3200, 0), ('K3200 bf16', 8704, 3200, 3200, H.GEMM_OUT_BF16),
                             ('M8704 N800 K3200 bf16', 8704, 800, 3200, H.GEMM_OUT_BF16), ('N3072 K808', 8704, 3072, 808, 0), ('N3328 K808', 8704, 3328, 808, 0)]:
    a = torch.randn(M, r8(K), device='cuda').to(torch.bfloat16)
    b = torch.randn(N, r8(K), device='cuda').to(torch.bfloat16)
    obf = bool(flags & H.GEMM_OUT_BF16)
    c = torch.zeros(M, r8(N), device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
    ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = flags & 0xFFFF
    if flags & BIAS:
        bias = torch.randn(N, device='cuda'); ep.bias = bias.data_ptr()
    if flags & (1 << 21):
        a = (a.float() * (torch.rand_like(a.float()) > 0.5)).to(torch.bfloat16)
    def run():
        lib.e2t_gemm_nt_bf16(a.data_ptr(), r8(K), b.data_ptr(), r8(K), c.data_ptr(), r8(N), M, N, r8(K), C.byref(ep), torch.cuda.current_stream().cuda_stream)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-24s M%5d N%5d K%5d  %8.1f us  %7.1f TFLOP/s  out %.0f MB -> %.2f TB/s' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, c.numel() * c.element_size() / 1e6, c.numel() * c.element_size() / us / 1e6), flush=True)
