"""Calibration only (NOT the product path): what the vendor BLAS reaches on the GEMM shapes of the cfg2 train step, as a
yardstick for the hand-written kernels (cdna_hip_programming.md 5.4 rule 10: a ceiling claim needs a known-good reference
measured on the same hardware).  torch.matmul on bf16 operands (hipBLASLt / rocBLAS underneath)."""
import torch, sys
dev = 'cuda:0'
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
shapes = [  # (name, form, M, N, K)
    ('dW_x   (K-major)', 'tn', 801, 3200, 8704), ('dW_h x2 (K-major)', 'tnb', 400, 1600, 8704), ('dW dec (K-major)', 'tn', 800, 3200, 2560),
    ('Gx      (NT)', 'nt', 8704, 3200, 832), ('dIn     (NT)', 'nt', 8704, 832, 3200), ('proj    (NT)', 'nt', 2560, 1806, 832),
    ('aux fwd (NT)', 'nt', 8704, 225, 832), ('conv    (NT)', 'nt', 8704, 100, 3136), ('square 4096', 'nt', 4096, 4096, 4096)]
for name, form, M, N, K in shapes:
    if form == 'nt':
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
        fn = lambda: torch.matmul(A, B.T)
        fl = 2.0 * M * N * K
    elif form == 'tn':
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        fn = lambda: torch.matmul(A.T, B)
        fl = 2.0 * M * N * K
    else:
        A = torch.randn(2, K, M, device=dev).bfloat16(); B = torch.randn(2, K, N, device=dev).bfloat16()
        fn = lambda: torch.bmm(A.transpose(1, 2), B)
        fl = 4.0 * M * N * K
    us = timeit(fn)
    print('%-20s M=%5d N=%5d K=%5d  %7.1f us  %7.1f TF' % (name, M, N, K, us, fl / us / 1e6))
