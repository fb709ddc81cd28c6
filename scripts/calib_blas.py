"""Calibration only (NOT the product path): what the vendor BLAS reaches on the GEMM shapes of the train steps of cfg2 / cfg4 /
cfg5, next to this library's kernels on the SAME shapes, the same box, isolated launches replayed from a hipGraph
(cdna_hip_programming.md 5.4 rule 10: a ceiling claim needs a known-good reference measured on the same hardware).
Vendor = torch.matmul on bf16 operands (hipBLASLt / rocBLAS underneath), which the product never calls.
    python scripts/calib_blas.py [cfg2|cfg4|cfg5|all]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
dev = 'cuda:0'


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


def r64(x):
    return (x + 63) // 64 * 64


SHAPES = {
    'cfg2': [('dW_x (K-major)', 'tn', 801, 3200, 8704, 1), ('dW_h x2 (K-major)', 'tn', 400, 1600, 8704, 2),
             ('dW dec (K-major)', 'tn', 800, 3200, 2560, 1), ('Gx (NT)', 'nt', 8704, 3200, 832, 1), ('Gx l0 (NT)', 'nt', 8704, 3200, 128, 1), ('dIn (NT)', 'nt', 8704, 832, 3200, 1),
             ('proj (NT)', 'nt', 2560, 1806, 832, 1), ('aux fwd (NT)', 'nt', 8704, 225, 832, 1), ('conv (NT)', 'nt', 8704, 100, 3136, 1),
             ('square 4096 (NT)', 'nt', 4096, 4096, 4096, 1), ('square 4096 (K-major)', 'tn', 4096, 4096, 4096, 1)],
    'cfg4': [('dW_x (K-major)', 'tn', 2049, 8192, 8704, 1), ('dW_x w/o bias row (K-major)', 'tn', 2048, 8192, 8704, 1),
             ('dW_h x2 (K-major)', 'tn', 1024, 4096, 8704, 2), ('dW_h dec (K-major)', 'tn', 2048, 8192, 2560, 1),
             ('dW_x l0 (K-major)', 'tn', 101, 8192, 8704, 1), ('proj dW (K-major)', 'tn', 1806, 2049, 2560, 1),
             ('Gx (NT)', 'nt', 8704, 8192, 2112, 1), ('dIn (NT)', 'nt', 8704, 2112, 8192, 1), ('Gx l0 (NT)', 'nt', 8704, 8192, 128, 1),
             ('Gx dec (NT)', 'nt', 2560, 8192, 192, 1), ('proj (NT)', 'nt', 2560, 1806, 2112, 1)],
    # the H_d = 2048 decoder's recurrent products of ONE time step as plain GEMMs (cfg4): what a GEMM + cell-kernel step would cost
    'dec4': [('dec rec fwd, one step (NT)', 'nt', 256, 8192, 2048, 1), ('dec rec bwd, one step (NT)', 'nt', 256, 2048, 8192, 1)],
    'cfg5': [('dW_x (K-major)', 'tn', 801, 3200, 42752, 1), ('dW_h x2 (K-major)', 'tn', 400, 1600, 42752, 2),
             ('conv dW (K-major)', 'tn', 12289, 100, 42752, 1), ('Gx (NT)', 'nt', 42752, 3200, 832, 1), ('dIn (NT)', 'nt', 42752, 832, 3200, 1),
             ('aux fwd (NT)', 'nt', 42752, 225, 832, 1)],
}

ws = torch.zeros(64 * 1024 * 1024, device=dev)           # 256 MiB of split-K slabs offered to the library


def ours(form, M, N, K, nb):
    ep = H.GemmEpilogue()
    ep.alpha = 1.0
    ep.splitk_ws, ep.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    if form == 'tn':
        lda, ldb = r64(M + 1), r64(N)
        A = torch.randn(nb, K, lda, device=dev).bfloat16(); Bm = torch.randn(nb, K, ldb, device=dev).bfloat16()
        Cm = torch.zeros(nb, M, N, device=dev)
        ep.flags = H.GEMM_SPLITK
        if nb > 1:
            ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = nb, K * lda, K * ldb, M * N
        fn = lambda: lib.e2t_gemm_tn_bf16(A.data_ptr(), lda, Bm.data_ptr(), ldb, Cm.data_ptr(), N, M, N, K, C.byref(ep), torch.cuda.current_stream().cuda_stream)
    else:
        Kp = r64(K)
        A = torch.randn(M, Kp, device=dev).bfloat16(); Bm = torch.randn(N, Kp, device=dev).bfloat16()
        Cm = torch.zeros(M, r64(N), device=dev, dtype=torch.bfloat16)
        ep.flags = H.GEMM_OUT_BF16
        fn = lambda: lib.e2t_gemm_nt_bf16(A.data_ptr(), Kp, Bm.data_ptr(), Kp, Cm.data_ptr(), r64(N), M, N, Kp, C.byref(ep), torch.cuda.current_stream().cuda_stream)
    tile, splits = C.c_int(0), C.c_int(0)
    lib.e2t_gemm_plan(int(form == 'tn'), M, N, K, C.byref(ep), C.byref(tile), C.byref(splits))
    us = timeit(fn)
    return us, tile.value, splits.value, (A, Bm, Cm, ep)


def vendor(form, M, N, K, nb):
    if form == 'nt':
        A = torch.randn(M, K, device=dev).bfloat16(); Bm = torch.randn(N, K, device=dev).bfloat16()
        return timeit(lambda: torch.matmul(A, Bm.T))
    A = torch.randn(nb, K, M, device=dev).bfloat16(); Bm = torch.randn(nb, K, N, device=dev).bfloat16()
    if nb == 1:
        return timeit(lambda: torch.matmul(A[0].T, Bm[0]))
    return timeit(lambda: torch.bmm(A.transpose(1, 2), Bm))


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
for cfg in (['cfg2', 'cfg4', 'cfg5'] if which == 'all' else [which]):
    print('--- %s' % cfg)
    for name, form, M, N, K, nb in SHAPES[cfg]:
        fl = 2.0 * M * N * K * nb
        uv = vendor(form, M, N, K, nb)
        uo, tile, splits, keep = ours(form, M, N, K, nb)
        del keep
        torch.cuda.empty_cache()
        print('%-30s M=%5d N=%5d K=%5d x%d  vendor %7.1f us %7.1f TF (%.3f)   ours %7.1f us %7.1f TF (%.3f)  tile %d splits %d'
              % (name, M, N, K, nb, uv, fl / uv / 1e6, fl / uv / 1e6 / 2500, uo, fl / uo / 1e6, fl / uo / 1e6 / 2500, tile, splits), flush=True)
