"""Micro-benchmark of e2t_gemm_nt_bf16 on the shapes of the cfg2 train step."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
def r8(x): return (x + 7) // 8 * 8
SHAPES = [  # (name, M, N, K, flags, out_bf16)
    ('Gx l1/l2   fp32 out', 8704, 3200, 800, 0),
    ('Gx l0      fp32 out', 8704, 3200, 104, 0),
    ('dIn        fp32 out', 8704, 800, 3200, 0),
    ('dWx  splitK        ', 801, 3200, 8704, H.GEMM_SPLITK),
    ('dWh  splitK        ', 400, 1600, 8704, H.GEMM_SPLITK),
    ('conv fwd   bf16 out', 8704, 100, 3072, H.GEMM_OUT_BF16),
    ('conv dW splitK     ', 3073, 100, 8704, H.GEMM_SPLITK),
    ('logits     fp32 out', 2560, 1806, 800, 0),
    ('dWp  splitK        ', 1806, 801, 2560, H.GEMM_SPLITK),
    ('square 4096        ', 4096, 4096, 4096, 0),
]
st = torch.cuda.current_stream().cuda_stream
wsbuf = torch.zeros(16 * 1024 * 1024, device='cuda')
for name, M, N, K, flags in SHAPES:
    a = torch.randn(M, r8(K), device='cuda').to(torch.bfloat16)
    b = torch.randn(N, r8(K), device='cuda').to(torch.bfloat16)
    obf = bool(flags & H.GEMM_OUT_BF16)
    c = torch.zeros(M, r8(N), device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
    ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = flags
    ep.splitk_ws, ep.splitk_ws_bytes = wsbuf.data_ptr(), wsbuf.numel() * 4
    def run():
        lib.e2t_gemm_nt_bf16(a.data_ptr(), r8(K), b.data_ptr(), r8(K), c.data_ptr(), r8(N), M, N, r8(K), C.byref(ep), st)
    run(); torch.cuda.synchronize()
    if True:
        ref = a.float() @ b.float().T
        err = (c.float()[:, :N] - ref).abs().max().item() / ref.abs().max().item()
    else:
        err = float('nan')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print('%-22s M%5d N%5d K%5d  %8.1f us  %7.1f TFLOP/s  relerr %.1e' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, err), flush=True)
