"""What each epilogue feature of k_gemm_nt costs on a small product (aux head hidden layer 8704 x 225 x 832; conv 8704 x 100 x 3136)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
dev = 'cuda:0'
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (M, N, K) in [(8704, 225, 832), (8704, 100, 3136), (8704, 832, 3200), (2560, 1806, 832)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    ldc = (N + 63) // 64 * 64
    Cf = torch.zeros(M, ldc, device=dev); Cb = torch.zeros(M, ldc, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev); lens = torch.full((256,), 34, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = torch.zeros(16 << 20, device=dev)
    mask = (torch.rand(M, ldc, device=dev) > 0.3).to(torch.bfloat16)
    def run(flags=0, bias_=False, drop=False, rows=False, bf16=False, msk=False, ws=False):
        ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = flags | (H.GEMM_OUT_BF16 if bf16 else 0)
        if bias_: ep.bias = bias.data_ptr()
        if drop:
            ep.flags |= H.GEMM_DROPOUT; ep.drop_rate, ep.drop_seed, ep.drop_step, ep.drop_stream, ep.drop_ld = 0.1, 5, step.data_ptr(), 3, N
        if rows: ep.row_lens, ep.rows_per_step = lens.data_ptr(), 256
        if msk: ep.relu_bwd_src, ep.ld_relu_bwd_src = mask.data_ptr(), ldc
        if ws: ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        out = Cb if bf16 else Cf
        return timeit(lambda: lib.e2t_gemm_nt_bf16(A.data_ptr(), K, B.data_ptr(), K, out.data_ptr(), ldc, M, N, K, C.byref(ep), torch.cuda.current_stream().cuda_stream))
    print('M=%d N=%d K=%d: plain fp32 %.1f | bf16 %.1f | +bias+relu %.1f | +rows %.1f | +dropout %.1f | all %.1f | mask (relu bwd) bf16 %.1f | plain+ws(split) %.1f us' % (
        M, N, K, run(), run(bf16=True), run(H.GEMM_RELU, bias_=True, bf16=True), run(rows=True, bf16=True), run(drop=True, bf16=True),
        run(H.GEMM_RELU, bias_=True, drop=True, rows=True, bf16=True), run(msk=True, bf16=True), run(ws=True)))
