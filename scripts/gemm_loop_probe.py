"""Time per K tile of the K-major GEMM main loop on a shape with exactly 1 or 2 workgroups per CU (diagnostics).
usage: python scripts/gemm_loop_probe.py [wgs_per_cu=1|2] ; env E2T_GEMM_DBG / E2T_TN_PIPE / E2T_GEMM_ORDER select the variant"""
import sys, ctypes as C
import torch
sys.path.insert(0, '.')
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
from ecog2txt_amd.hip_lib import lib, GemmEpilogue

per = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = len(sys.argv) > 2 and sys.argv[2] == "nt"
zero = len(sys.argv) > 3 and sys.argv[3] == "zero"
M, N = 2048, 2048 * per
dev = 'cuda:0'
res = []
for K in (2048, 8192):
    A = (torch.randn(M, K, device=dev) if nt else torch.randn(K, M, device=dev)).bfloat16()
    B = (torch.randn(N, K, device=dev) if nt else torch.randn(K, N, device=dev)).bfloat16()
    if zero: A.zero_(); B.zero_()
    Cc = torch.zeros(M, N, device=dev)
    ep = GemmEpilogue(); ep.alpha = 1.0
    st = torch.cuda.current_stream().cuda_stream
    def go():
        if nt: lib.e2t_gemm_nt_bf16(A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, M, N, K, C.byref(ep), st)
        else: lib.e2t_gemm_tn_bf16(A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, M, N, K, C.byref(ep), st)
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    res.append((K, e0.elapsed_time(e1) / 20 * 1e3))
(k0, t0), (k1, t1) = res
print('M=%d N=%d: K=%d %.1f us, K=%d %.1f us -> %.3f us per 64-deep K tile, fixed %.1f us; main-loop rate %.0f TF' % (
    M, N, k0, t0, k1, t1, (t1 - t0) / ((k1 - k0) / 64), t0 - (t1 - t0) / (k1 - k0) * k0, 2.0 * M * N * (k1 - k0) / (t1 - t0) / 1e6))
