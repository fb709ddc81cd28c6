"""Kernel-level parity tests (call through the C ABI, compare with NumPy/oracle)."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from oracle.bf16 import round_bf16, to_bf16_bits, from_bf16_bits
from oracle.philox import keep_mask
from oracle import seq2seq as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hl():
    from ecog2txt_amd import hip_lib
    hip_lib.load()
    return hip_lib


def dev_bf16(a):
    """numpy float -> cuda bf16 tensor holding round_bf16(a)."""
    bits = to_bf16_bits(a).astype(np.int16)
    return torch.from_numpy(bits).cuda().view(torch.bfloat16)


def host(t):
    if t.dtype == torch.bfloat16:
        return from_bf16_bits(t.view(torch.int16).cpu().numpy().view(np.uint16))
    return t.cpu().numpy().astype(np.float64)


def st():
    return torch.cuda.current_stream().cuda_stream


def r8(x):
    return (x + 7) // 8 * 8


@pytest.mark.parametrize('M,N,K', [(16, 16, 32), (128, 128, 64), (130, 70, 40), (300, 257, 136), (1, 50, 264),
                                   (513, 100, 3072)])
def test_gemm_nt_plain(hl, M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((N, K))            # asymmetric operands catch transposed outputs
    a, b = dev_bf16(A), dev_bf16(Bm)
    c = torch.full((M, N), 7.0, dtype=torch.float32, device='cuda')
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N, K, None, st())
    torch.cuda.synchronize()
    want = round_bf16(A) @ round_bf16(Bm).T
    np.testing.assert_allclose(host(c), want, rtol=1e-5, atol=1e-4 * np.sqrt(K))


@pytest.mark.parametrize('N', [52, 53, 225, 50])
def test_gemm_epilogues(hl, N):
    """(N % 4 != 0: a lane's four columns straddle two Philox blocks in three rows out of four -- the two-block path of the
    dropout epilogue; N = 225 is the auxiliary head's hidden width.)"""
    rng = np.random.default_rng(3)
    M, K, rowsB = 96, 72, 8
    A, Bm = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    bias = rng.standard_normal(N)
    lens = rng.integers(0, M // rowsB + 1, size=rowsB)
    a, b = dev_bf16(A), dev_bf16(Bm)
    bt = torch.tensor(bias, dtype=torch.float32, device='cuda')
    lt = torch.tensor(lens, dtype=torch.int32, device='cuda')
    step = torch.tensor([5], dtype=torch.int32, device='cuda')
    ldc = r8(N)
    out = torch.zeros(M, ldc, dtype=torch.bfloat16, device='cuda')
    ep = hl.GemmEpilogue()
    ep.bias, ep.alpha = bt.data_ptr(), 1.0
    ep.flags = hl.GEMM_RELU | hl.GEMM_OUT_BF16 | hl.GEMM_DROPOUT
    ep.drop_rate, ep.drop_seed, ep.drop_step, ep.drop_stream, ep.drop_ld = 0.25, 1000, step.data_ptr(), 9, N
    ep.row_lens, ep.rows_per_step = lt.data_ptr(), rowsB
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), ldc, M, N, K, C.byref(ep), st())
    torch.cuda.synchronize()
    ref = np.maximum(round_bf16(A) @ round_bf16(Bm).T + bias, 0.0)
    ref = ref * keep_mask((M, N), 0.25, 1005, 9) / 0.75
    valid = (np.arange(M) // rowsB) < lens[np.arange(M) % rowsB]
    ref = round_bf16(ref * valid[:, None])
    got = host(out)[:, :N]
    # bf16 outputs: allow one ulp where fp32 vs fp64 accumulation straddles a rounding boundary
    np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=1e-6)
    assert (got[~valid] == 0).all()
    assert np.all(host(out)[:, N:] == 0)
    # accumulate + relu-backward mask + alpha
    c0 = rng.standard_normal((M, N))
    ct = torch.tensor(c0, dtype=torch.float32, device='cuda')
    ep2 = hl.GemmEpilogue()
    ep2.alpha, ep2.flags = 0.5, hl.GEMM_ACCUMULATE
    ep2.relu_bwd_src, ep2.ld_relu_bwd_src = out.data_ptr(), ldc
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, ct.data_ptr(), N, M, N, K, C.byref(ep2), st())
    torch.cuda.synchronize()
    want = c0 + 0.5 * (round_bf16(A) @ round_bf16(Bm).T) * (ref != 0)
    np.testing.assert_allclose(host(ct), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('M,N,K', [(16, 16, 64), (128, 128, 256), (130, 70, 200), (801, 400, 1000), (100, 3073, 333), (37, 9, 5)])
def test_gemm_tn_k_major_operands(hl, M, N, K):
    """C = A^T . B with both operands K-major (the weight-gradient shape): ragged K, M, N; with and without split-K."""
    rng = np.random.default_rng(M + 3 * N + K)
    lda, ldb = r8(M) + 8, r8(N)
    A = np.zeros((K, lda)); A[:, :M] = rng.standard_normal((K, M))
    Bm = np.zeros((K, ldb)); Bm[:, :N] = rng.standard_normal((K, N))
    A[:, M:] = 7.0                                           # columns beyond M must never reach the output
    a, b = dev_bf16(A), dev_bf16(Bm)
    want = round_bf16(A[:, :M]).T @ round_bf16(Bm[:, :N])
    wsb = torch.zeros(4 * 1024 * 1024, dtype=torch.float32, device='cuda')
    for split in (False, True):
        c = torch.full((M, r8(N)), 7.0, dtype=torch.float32, device='cuda')
        ep = hl.GemmEpilogue(); ep.alpha = 1.0
        if split:
            ep.flags = hl.GEMM_SPLITK
            ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), r8(N), M, N, K, C.byref(ep), st())
        torch.cuda.synchronize()
        np.testing.assert_allclose(host(c)[:, :N], want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
        assert np.all(host(c)[:, N:] == 7.0)


@pytest.mark.parametrize('M,N,K,want_splits', [(130, 132, 2048, 2), (130, 132, 4096, 4), (128, 128, 8704, 8), (130, 130, 2048, 2), (5, 8, 2048, 2)])
def test_splitk_reduction_group_counts(hl, M, N, K, want_splits):
    """The reduction of a weight gradient (nothing between the sum and the fp32 store, N % 4 == 0) takes 4 / 2 / 1 column groups per
    thread for <= 2 / <= 4 / more slabs, with a ragged last group; any other product goes through the full epilogue, one group
    per thread.  Same sums in the same split order either way."""
    rng = np.random.default_rng(M + N + K)
    lda, ldb = r8(M) + 8, r8(N)
    A = np.zeros((K, lda)); A[:, :M] = rng.standard_normal((K, M))
    Bm = np.zeros((K, ldb)); Bm[:, :N] = rng.standard_normal((K, N))
    a, b = dev_bf16(A), dev_bf16(Bm)
    want = round_bf16(A[:, :M]).T @ round_bf16(Bm[:, :N])
    wsb = torch.zeros(4 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ldc = r8(N) + 4
    c = torch.full((M, ldc), 7.0, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 1.0
    ep.flags = hl.GEMM_SPLITK
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    tile, splits = C.c_int(0), C.c_int(0)
    hl.lib.e2t_gemm_plan(1, M, N, K, C.byref(ep), C.byref(tile), C.byref(splits))
    assert splits.value == want_splits
    hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc, M, N, K, C.byref(ep), st())
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(c)[:, :N], want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
    assert np.all(host(c)[:, N:] == 7.0)


@pytest.mark.parametrize('split', [False, True])
def test_gemm_batched_products_in_one_launch(hl, split):
    """epilogue.batch: z-th product reads A + z*a_stride, B + z*b_stride and writes C + z*c_stride (the two directions
    of a recurrent weight gradient: shifted rows of one activation array, column blocks of one gradient array)."""
    rng = np.random.default_rng(11)
    M, N, K, nb = 72, 136, 1100, 2
    lda, ldb = 2 * r8(M) + 8, nb * r8(N)
    A = rng.standard_normal((K + 16, lda)); Bm = rng.standard_normal((K, ldb))
    a, b = dev_bf16(A), dev_bf16(Bm)
    a_stride, b_stride, c_stride = 5 * lda + r8(M), r8(N), M * r8(N) + 40         # rows AND columns shift between the products
    c = torch.full((nb * (M * r8(N) + 40),), 7.0, dtype=torch.float32, device='cuda')
    wsb = torch.zeros(4 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 1.0
    ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = nb, a_stride, b_stride, c_stride
    if split:
        ep.flags = hl.GEMM_SPLITK
        ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), r8(N), M, N, K, C.byref(ep), st())
    torch.cuda.synchronize()
    out = host(c)
    Ar, Br = round_bf16(A), round_bf16(Bm)
    for z in range(nb):
        want = Ar[5 * z:5 * z + K, z * r8(M):z * r8(M) + M].T @ Br[:, z * r8(N):z * r8(N) + N]
        got = out[z * c_stride:z * c_stride + M * r8(N)].reshape(M, r8(N))
        np.testing.assert_allclose(got[:, :N], want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
    assert np.all(out[M * r8(N):c_stride] == 7.0)


@pytest.mark.parametrize('ws_floats', [8 * 1024 * 1024, 600 * 1024, 1024])
def test_gemm_tn_group_one_launch_for_several_products(hl, ws_floats):
    """e2t_gemm_tn_group_bf16: several K-major products (different shapes, one of them batched, one with its last column
    diverted, ragged edges) in ONE launch equal the products computed one by one -- with the K splits the library picks for
    the group, with a workspace that only admits shallow splits, and with one too small for any split."""
    rng = np.random.default_rng(17)
    wsb = torch.zeros(ws_floats, dtype=torch.float32, device='cuda')
    shapes = [(801, 400, 2300, 1), (72, 136, 1100, 2), (130, 70, 2000, 1), (37, 9, 70, 1), (256, 384, 4096, 1)]
    calls = (hl.GemmCall * len(shapes))()
    keep, checks = [], []
    for i, (M, N, K, nb) in enumerate(shapes):
        lda, ldb = nb * r8(M) + 8, nb * r8(N)
        A = rng.standard_normal((K + 16, lda)); Bm = rng.standard_normal((K, ldb))
        a, b = dev_bf16(A), dev_bf16(Bm)
        ep = hl.GemmEpilogue(); ep.alpha = 1.0; ep.flags = hl.GEMM_SPLITK
        ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        a_stride, b_stride, c_stride = 5 * lda + r8(M), r8(N), M * r8(N) + 40
        if nb > 1:
            ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = nb, a_stride, b_stride, c_stride
        c = torch.full((nb * (M * r8(N) + 40),), 7.0, dtype=torch.float32, device='cuda')
        lc = None
        if i == 2:                   # last column of the product -> its own vector (the bias gradient of a transposed kernel)
            lc = torch.full((M,), 7.0, dtype=torch.float32, device='cuda')
            ep.last_col_out = lc.data_ptr()
        calls[i].A, calls[i].lda, calls[i].B, calls[i].ldb = a.data_ptr(), lda, b.data_ptr(), ldb
        calls[i].C, calls[i].ldc, calls[i].M, calls[i].N, calls[i].K = c.data_ptr(), r8(N), M, N, K
        calls[i].ep = C.pointer(ep)
        keep.append((a, b, ep))
        checks.append((A, Bm, c, lc, M, N, K, nb, a_stride, b_stride, c_stride))
    hl.lib.e2t_gemm_tn_group_bf16(len(shapes), calls, st())
    torch.cuda.synchronize()
    for (A, Bm, c, lc, M, N, K, nb, a_stride, b_stride, c_stride) in checks:
        out = host(c)
        Ar, Br = round_bf16(A), round_bf16(Bm)
        for z in range(nb):
            if nb > 1:
                want = Ar[5 * z:5 * z + K, z * r8(M):z * r8(M) + M].T @ Br[:, z * r8(N):z * r8(N) + N]
            else:
                want = Ar[:K, :M].T @ Br[:, :N]
            got = out[z * c_stride:z * c_stride + M * r8(N)].reshape(M, r8(N))
            if lc is not None:
                np.testing.assert_allclose(host(lc), want[:, N - 1], rtol=1e-5, atol=1e-4 * np.sqrt(K))
                np.testing.assert_allclose(got[:, :N - 1], want[:, :N - 1], rtol=1e-5, atol=1e-4 * np.sqrt(K))
            else:
                np.testing.assert_allclose(got[:, :N], want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
            assert np.all(got[:, N:] == 7.0)


def test_gemm_splitk_runs_the_full_epilogue(hl):
    """Few output tiles + long K: the library splits K on its own when a workspace is offered, and the reduction
    applies the same bias / ReLU / dropout / row mask / bf16 epilogue (same Philox mask) as the direct store."""
    rng = np.random.default_rng(5)
    M, N, K, rowsB = 200, 100, 2048, 8
    A, Bm = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    bias = rng.standard_normal(N)
    lens = rng.integers(0, M // rowsB + 1, size=rowsB)
    a, b = dev_bf16(A), dev_bf16(Bm)
    bt = torch.tensor(bias, dtype=torch.float32, device='cuda')
    lt = torch.tensor(lens, dtype=torch.int32, device='cuda')
    step = torch.tensor([2], dtype=torch.int32, device='cuda')
    wsb = torch.zeros(4 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ldc = r8(N)
    outs = []
    for use_ws in (False, True):
        out = torch.zeros(M, ldc, dtype=torch.bfloat16, device='cuda')
        ep = hl.GemmEpilogue()
        ep.bias, ep.alpha = bt.data_ptr(), 1.0
        ep.flags = hl.GEMM_RELU | hl.GEMM_OUT_BF16 | hl.GEMM_DROPOUT
        ep.drop_rate, ep.drop_seed, ep.drop_step, ep.drop_stream, ep.drop_ld = 0.25, 77, step.data_ptr(), 3, N
        ep.row_lens, ep.rows_per_step = lt.data_ptr(), rowsB
        if use_ws:
            ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), ldc, M, N, K, C.byref(ep), st())
        torch.cuda.synchronize()
        outs.append(host(out)[:, :N])
    assert float(wsb.abs().max()) > 0, 'the split-K path must have been taken'
    ref = np.maximum(round_bf16(A) @ round_bf16(Bm).T + bias, 0.0)
    ref = ref * keep_mask((M, N), 0.25, 79, 3) / 0.75
    valid = (np.arange(M) // rowsB) < lens[np.arange(M) % rowsB]
    ref = round_bf16(ref * valid[:, None])
    for got in outs:
        np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=1e-5)
    assert ((outs[0] == 0) == (outs[1] == 0)).all()            # identical masks
    # fp32 accumulate through the split path, with the last column diverted (bias-gradient column of the dW GEMMs)
    c0 = rng.standard_normal((M, N - 1))
    ct = torch.tensor(c0, dtype=torch.float32, device='cuda')
    lastc = torch.zeros(M, dtype=torch.float32, device='cuda')
    ep2 = hl.GemmEpilogue()
    ep2.alpha, ep2.flags = 0.5, hl.GEMM_ACCUMULATE | hl.GEMM_SPLITK
    ep2.last_col_out = lastc.data_ptr()
    ep2.splitk_ws, ep2.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, ct.data_ptr(), N - 1, M, N, K, C.byref(ep2), st())
    torch.cuda.synchronize()
    full = 0.5 * (round_bf16(A) @ round_bf16(Bm).T)
    np.testing.assert_allclose(host(ct), c0 + full[:, :N - 1], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(host(lastc), full[:, N - 1], rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize('R,Cc', [(5, 3), (64, 64), (100, 37), (257, 130)])
def test_transpose(hl, R, Cc):
    rng = np.random.default_rng(R)
    x = rng.standard_normal((R, r8(Cc)))
    xt = dev_bf16(x)
    ldo = r8(R) + 8
    out = torch.full((Cc + 1, ldo), 1.0, dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_transpose_bf16(xt.data_ptr(), r8(Cc), R, Cc, out.data_ptr(), ldo, st())
    torch.cuda.synchronize()
    got = host(out)
    np.testing.assert_array_equal(got[:Cc, :R], round_bf16(x)[:, :Cc].T)
    assert (got[:Cc, R:] == 0).all()          # K padding zero-filled
    assert (got[Cc] == 1).all()               # ones row untouched


@pytest.mark.parametrize('C_', [8, 6])
def test_lengths_and_conv_pack(hl, C_):
    rng = np.random.default_rng(C_)
    B, T, N = 7, 23, 4
    X = np.abs(rng.standard_normal((B, T, C_))) + 0.1
    lens = np.array([23, 1, 0, 10, 22, 4, 5])
    for b in range(B):
        X[b, lens[b]:] = 0
    xt = torch.tensor(X, dtype=torch.float32, device='cuda')
    lt = torch.zeros(B, dtype=torch.int32, device='cuda')
    ld = torch.zeros(B, dtype=torch.int32, device='cuda')
    hl.lib.e2t_seq_lengths_f32(xt.data_ptr(), B, T, C_, N, lt.data_ptr(), ld.data_ptr(), st())
    S = -(-T // N)
    K8 = r8(N * C_)
    A = torch.full((S * B, K8), 3.0, dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_conv_pack(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, A.data_ptr(), K8, st())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lt.cpu().numpy(), lens)
    np.testing.assert_array_equal(ld.cpu().numpy(), -(-lens // N))
    Xr = O.reverse_time_major(X.astype(np.float32).astype(np.float64), lens)
    Xp = np.zeros((S * N, B, C_))
    Xp[:T] = Xr
    want = round_bf16(Xp.reshape(S, N, B, C_).transpose(0, 2, 1, 3).reshape(S * B, N * C_))
    got = host(A)
    np.testing.assert_array_equal(got[:, :N * C_], want)
    assert (got[:, N * C_:] == 0).all()


@pytest.mark.parametrize('C_,T', [(256, 400), (1024, 150), (64, 33), (36, 70), (6, 23)])
def test_lengths_from_the_tail_equal_the_row_count_for_end_padded_batches(hl, C_, T):
    """e2t_seq_lengths_tail_f32 (searches the padding from the end, 32 rows per pass) against the full non-zero-row count
    and NumPy, on lengths that hit the pass boundaries; a row whose only non-zero sits in the last channel counts."""
    rng = np.random.default_rng(C_ + T)
    B, N = 13, 12
    lens = np.array([T, 0, 1, T - 1, T - 31, T - 32, T - 33, max(T - 64, 0), 2, 31, 32, min(33, T), T // 2])
    lens = np.clip(lens, 0, T)
    X = np.zeros((B, T, C_), np.float32)
    for b in range(B):
        X[b, :lens[b]] = np.abs(rng.standard_normal((lens[b], C_))) + 0.1
        if lens[b] > 0:
            X[b, lens[b] - 1] = 0
            X[b, lens[b] - 1, C_ - 1] = 1e-3
    xt = torch.tensor(X, device='cuda')
    out = {}
    for name in ('e2t_seq_lengths_f32', 'e2t_seq_lengths_tail_f32'):
        lt = torch.full((B,), -7, dtype=torch.int32, device='cuda')
        ld = torch.full((B,), -7, dtype=torch.int32, device='cuda')
        getattr(hl.lib, name)(xt.data_ptr(), B, T, C_, N, lt.data_ptr(), ld.data_ptr(), st())
        torch.cuda.synchronize()
        out[name] = (lt.cpu().numpy(), ld.cpu().numpy())
    for lt, ld in out.values():
        np.testing.assert_array_equal(lt, lens)
        np.testing.assert_array_equal(ld, -(-lens // N))
    np.testing.assert_array_equal(O.sequence_lengths(X), lens)


def test_softmax_ce(hl):
    rng = np.random.default_rng(0)
    B, L, V = 6, 5, 1806
    M = B * L
    logits = 3 * rng.standard_normal((M, V))
    tgt = rng.integers(0, V, size=M)
    lens = np.array([5, 1, 0, 3, 5, 2])
    lg = torch.tensor(logits, dtype=torch.float32, device='cuda')
    tg = torch.tensor(tgt, dtype=torch.int32, device='cuda')
    ln = torch.tensor(lens, dtype=torch.int32, device='cuda')
    ntok = torch.tensor([int(lens.sum())], dtype=torch.int32, device='cuda')
    rl = torch.zeros(M, dtype=torch.float32, device='cuda'); cr = torch.zeros(M, dtype=torch.float32, device='cuda')
    pr = torch.zeros(M, dtype=torch.int32, device='cuda')
    dl = torch.zeros(M, r8(V), dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_softmax_ce(lg.data_ptr(), V, M, V, tg.data_ptr(), ln.data_ptr(), B, ntok.data_ptr(), 0.7, rl.data_ptr(),
                          pr.data_ptr(), cr.data_ptr(), dl.data_ptr(), r8(V), st())
    torch.cuda.synchronize()
    l32 = logits.astype(np.float32).astype(np.float64)
    mx = l32.max(1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(l32 - mx).sum(1))
    valid = (np.arange(M) // B) < lens[np.arange(M) % B]
    np.testing.assert_allclose(host(rl), (lse - l32[np.arange(M), tgt]) * valid, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(pr.cpu().numpy(), l32.argmax(1))
    p = np.exp(l32 - lse[:, None])
    oh = np.zeros_like(p); oh[np.arange(M), tgt] = 1
    want = (p - oh) * valid[:, None] * 0.7 / lens.sum()
    np.testing.assert_allclose(host(dl)[:, :V], want, rtol=2 ** -7, atol=1e-7)


@pytest.mark.parametrize('n,off', [(1000, 0), (1003, 1), (2, 3), (4099, 2), (5, 0)])
def test_adam_ema_matches_oracle(hl, n, off):
    """(off: element offset of the range inside its buffers -- the 16-B body of the kernel is framed by a scalar head and tail)"""
    rng = np.random.default_rng(1)
    p0, g1, g2 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    P, state = {'w': p0.copy()}, {}
    for g in (g1, g2):
        O.adam_ema_step(P, {'w': g}, state, lr=1e-2)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    pad = lambda a: np.concatenate([np.full(off, 9.0), a, np.full(7, 9.0)])
    p, m, v, ema = t(pad(p0)), t(pad(np.zeros(n))), t(pad(np.zeros(n))), t(pad(p0))
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    h = hl.AdamHyper(1e-2, 0.9, 0.999, 1e-8, 0.99, 1.0)
    o = 4 * off
    for g in (g1, g2):
        hl.lib.e2t_inc_step(step.data_ptr(), None, st())
        gt = t(pad(g))
        hl.lib.e2t_adam_ema_step(p.data_ptr() + o, gt.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, ema.data_ptr() + o, n,
                                 step.data_ptr(), C.byref(h), st())
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(p)[off:off + n], P['w'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(host(ema)[off:off + n], state['ema']['w'], rtol=2e-5, atol=1e-6)
    for a in (p, ema):                          # nothing outside the range is touched
        assert np.all(host(a)[:off] == 9.0) and np.all(host(a)[off + n:] == 9.0)
    assert np.all(host(m)[:off] == 9.0) and np.all(host(m)[off + n:] == 9.0)


@pytest.mark.parametrize('M,N,K', [(101, 200, 8704), (401, 130, 2560), (5, 14, 1040), (130, 70, 40)])
def test_gemm_splitk_and_bias_column(hl, M, N, K):
    """Weight-gradient form: split-K partials meet by atomics in a zeroed C; the last column of the
    product (B's ones row) goes to a separate bias-gradient vector."""
    rng = np.random.default_rng(K + M)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((N + 1, K))
    Bm[N] = 1.0
    a, b = dev_bf16(A), dev_bf16(Bm)
    c = torch.full((M, N), 3.0, dtype=torch.float32, device='cuda')      # must be overwritten, not accumulated
    colv = torch.full((M,), 3.0, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue()
    ep.alpha, ep.flags, ep.last_col_out = 1.0, hl.GEMM_SPLITK, colv.data_ptr()
    wsb = torch.zeros(8 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N + 1, K, C.byref(ep), st())
    torch.cuda.synchronize()
    want = round_bf16(A) @ round_bf16(Bm).T
    np.testing.assert_allclose(host(c), want[:, :N], rtol=1e-5, atol=2e-4 * np.sqrt(K))
    np.testing.assert_allclose(host(colv), want[:, N], rtol=1e-5, atol=2e-4 * np.sqrt(K))


def test_pack_batch_all_kinds(hl):
    """Every descriptor kind of e2t_pack_batch against NumPy: strided cast (scalar and 16-B), transposing cast (scalar
    and 16-B), MFMA fragment images (k-strided, k-contiguous, n-contiguous via LDS, gate-interleaved four-in-one)."""
    rng = np.random.default_rng(21)
    base = rng.standard_normal(400000).astype(np.float32)
    bt = torch.tensor(base, device='cuda')
    U = hl.PACK_UNITS

    def frag_image(Bn, Nn, Kk):                       # [NT][KB][64 lanes][8]: lane (n = l&15, kgroup = l>>4)
        NT, KB = -(-Nn // 16), -(-Kk // 32)
        out = np.zeros((NT, KB, 64, 8))
        for nt in range(NT):
            for kb in range(KB):
                for l in range(64):
                    n, k0 = nt * 16 + (l & 15), kb * 32 + (l >> 4) * 8
                    for j in range(8):
                        if n < Nn and k0 + j < Kk:
                            out[nt, kb, l, j] = Bn[n, k0 + j]
        return round_bf16(out)

    cases = []          # (kind, src_off, s0, s1, d0, d1, ld, units, expected builder)
    # 0 / 3: dst[r][c] = src[r*s0 + c*s1]
    cases.append((0, 3, 37, 1, 20, 30, 32, 20 * 1, lambda off, s0, s1, d0, d1, ld: ('mat', np.array([[base[off + r * s0 + c * s1] for c in range(d1)] for r in range(d0)]))))
    cases.append((3, 8, 1200, 1, 20, 1100, 1104, 20 * 2, cases[0][8]))
    # 2 / 4: same mapping, source contiguous along r
    cases.append((2, 5, 1, 70, 66, 50, 56, 2 * 1, cases[0][8]))
    cases.append((4, 16, 1, 72, 68, 52, 56, 2 * 1, cases[0][8]))
    # 1: fragment image of Bn[n][k] = src[n*s0 + k*s1]
    fr = lambda off, s0, s1, d0, d1, ld: ('frag', np.array([[base[off + n * s0 + k * s1] for k in range(d1)] for n in range(d0)]))
    cases.append((1, 7, 3, 61, 20, 40, 2, -(-(2 * 2) // 4), fr))                      # scalar path
    cases.append((1, 12, 100, 1, 36, 72, 3, -(-(3 * 3) // 4), fr))                    # k-contiguous: 16-B loads
    cases.append((5, 20, 1, 44, 40, 70, 3, -(-3 // 4) * -(-3 // 2), fr))                # n-contiguous via LDS: 2 k-blocks per workgroup, odd count
    descs = (hl.PackDesc * (len(cases) + 1))()
    outs, nblk = [], 0
    for i, (kind, off, s0, s1, d0, d1, ld, units, _) in enumerate(cases):
        n_out = (-(-d0 // 16) * ld * 512) if kind in (1, 5) else d0 * ld
        o = torch.full((n_out + 64,), 7.0, dtype=torch.bfloat16, device='cuda')
        outs.append(o)
        d = descs[i]
        d.kind, d.first_block, d.src_off, d.s0, d.s1, d.d0, d.d1, d.ld, d.dst = kind, nblk, off, s0, s1, d0, d1, ld, o.data_ptr()
        nblk += -(-units // U) if kind in (1, 3) else units
    # 6: four gate images from a gate-interleaved source [k][unit*4 + gate]
    Hh, Kk, s1 = 24, 72, 4 * 24                      # 3 k-blocks: two per workgroup, the last workgroup of a unit tile half empty
    off6 = 100000
    UT, KB = -(-Hh // 16), -(-Kk // 32)
    o6 = torch.full((4 * UT * KB * 512 + 64,), 7.0, dtype=torch.bfloat16, device='cuda')
    d = descs[len(cases)]
    d.kind, d.first_block, d.src_off, d.s0, d.s1, d.d0, d.d1, d.ld, d.dst = 6, nblk, off6, 4, s1, Hh, Kk, KB, o6.data_ptr()
    nblk += UT * -(-KB // 2)
    dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to('cuda')
    hl.lib.e2t_pack_batch(dev.data_ptr(), len(cases) + 1, nblk, bt.data_ptr(), st())
    torch.cuda.synchronize()
    for (kind, off, s0, s1_, d0, d1, ld, units, build), o in zip(cases, outs):
        what, ref = build(off, s0, s1_, d0, d1, ld)
        got = o.float().cpu().numpy()
        if what == 'mat':
            g = got[:d0 * ld].reshape(d0, ld)
            np.testing.assert_array_equal(g[:, :d1], round_bf16(ref), err_msg='kind %d' % kind)
            assert np.all(g[:, d1:] == 7.0) and np.all(got[d0 * ld:] == 7.0)
        else:
            want = frag_image(ref, d0, d1)
            np.testing.assert_array_equal(got[:want.size].reshape(want.shape), want, err_msg='kind %d' % kind)
    got6 = o6.float().cpu().numpy()[:4 * UT * KB * 512].reshape(4, UT, KB, 64, 8)
    for g in range(4):
        Bn = np.array([[base[off6 + (n * 4 + g) + k * s1] for k in range(Kk)] for n in range(Hh)])
        np.testing.assert_array_equal(got6[g], frag_image(Bn, Hh, Kk), err_msg='kind 6 gate %d' % g)


@pytest.mark.parametrize('row_words', [400 * 256, 10, 7, 1030])
def test_gather_rows_assembles_a_batch_with_padding_rows(hl, row_words):
    """e2t_gather_rows_u32: dst row r = src row idx[r]; idx < 0 and rows beyond n are zero (padding utterances).
    Bit-exact (a copy); vector (16-B) and scalar row sizes."""
    rng = np.random.default_rng(row_words)
    n_src, n, rows_out = 37, 19, 24
    src = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(n_src, row_words), dtype=np.int64).astype(np.int32)
    idx = rng.integers(0, n_src, size=n).astype(np.int32)
    idx[3] = -1
    d_src, d_idx = torch.from_numpy(src).cuda(), torch.from_numpy(idx).cuda()
    dst = torch.full((rows_out, row_words), 123, dtype=torch.int32, device='cuda')
    hl.lib.e2t_gather_rows_u32(d_src.data_ptr(), d_idx.data_ptr(), n, rows_out, row_words, dst.data_ptr(), st())
    torch.cuda.synchronize()
    want = np.zeros((rows_out, row_words), np.int32)
    for r in range(n):
        if idx[r] >= 0:
            want[r] = src[idx[r]]
    np.testing.assert_array_equal(dst.cpu().numpy(), want)


def test_gather_rows_in_blocks(hl):
    """e2t_gather_rows_blocks_u32: the row selection of e2t_gather_rows_u32 applied to every block (bf16-staged inputs: block =
    decimated step, row = utterance)."""
    rng = np.random.default_rng(2)
    n_src, rows_out, n, rw, blocks = 23, 8, 6, 12, 5
    src = rng.integers(1, 2 ** 31, size=(blocks, n_src, rw), dtype=np.int64).astype(np.int32)
    idx = np.array([3, -1, 22, 0, 7, 7], np.int32)
    dst = torch.full((blocks, rows_out + 2, rw), 5, dtype=torch.int32, device='cuda')
    d_src, d_idx = torch.from_numpy(src).cuda(), torch.from_numpy(idx).cuda()
    hl.lib.e2t_gather_rows_blocks_u32(d_src.data_ptr(), d_idx.data_ptr(), n, rows_out, rw, blocks, n_src * rw, (rows_out + 2) * rw,
                                      dst.data_ptr(), st())
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    for t in range(blocks):
        for r in range(rows_out):
            want = src[t, idx[r]] if (r < n and idx[r] >= 0) else 0
            assert np.array_equal(got[t, r], np.broadcast_to(want, (rw,))), (t, r)
        assert np.all(got[t, rows_out:] == 5)


@pytest.mark.parametrize('C_,F,N,T,B', [(64, 100, 12, 50, 9), (256, 100, 12, 400, 40), (1024, 100, 12, 100, 24), (128, 128, 4, 30, 70), (64, 5, 3, 20, 3)])
def test_fused_conv_forward_matches_pack_plus_gemm(hl, C_, F, N, T, B):
    """e2t_conv_fwd_fused (one pass over the fp32 grid: reversal + im2row + bf16 rounding + GEMM + epilogue) against the
    NumPy restatement: bf16(x) . bf16(W) accumulated in fp64, + bias, ReLU, rows beyond the decimated length zeroed
    (oracle/seq2seq.py forward(), conv stage; trainers.py:808-818)."""
    rng = np.random.default_rng(C_ + F)
    S = -(-T // N)
    lens = rng.integers(1, T + 1, size=B)
    lens[0] = T
    if B > 2:
        lens[1] = 1
    X = (np.abs(rng.standard_normal((B, T, C_))) + 0.1).astype(np.float32)
    for b in range(B):
        X[b, lens[b]:] = 0
    K = N * C_
    ldw = (K + 1 + 63) // 64 * 64
    W = (rng.standard_normal((F, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(F).astype(np.float32) * 0.1
    WT = torch.zeros((F, ldw), dtype=torch.bfloat16, device='cuda')
    WT[:, :K] = dev_bf16(W)
    lde = (F + 63) // 64 * 64
    E = torch.full((S * B, lde), 7.0, dtype=torch.bfloat16, device='cuda')
    xt = torch.tensor(X, device='cuda')
    lt = torch.tensor(lens, dtype=torch.int32, device='cuda')
    ld = torch.tensor(-(-lens // N), dtype=torch.int32, device='cuda')
    bt = torch.tensor(bias, device='cuda')
    ep = hl.GemmEpilogue()
    ep.bias, ep.alpha, ep.flags = bt.data_ptr(), 1.0, hl.GEMM_RELU | hl.GEMM_OUT_BF16
    ep.row_lens, ep.rows_per_step = ld.data_ptr(), B
    assert hl.load().e2t_conv_fwd_fused_ok(C_, F) == 1
    wsb = torch.zeros(4 * S * B * F, dtype=torch.float32, device='cuda')
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    lda = (K + 1 + 7) // 8 * 8
    Ap = torch.full((S * B, lda), 3.0, dtype=torch.bfloat16, device='cuda')
    # (E2T_CONV_FWD=<K step>x<stages>[x<splits>] selects other pipeline shapes for the whole process: the same test covers them)
    hl.lib.e2t_conv_fwd_fused(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, WT.data_ptr(), ldw, E.data_ptr(), lde, F,
                              Ap.data_ptr(), lda, C.byref(ep), st())
    torch.cuda.synchronize()
    Xr = O.reverse_time_major(X.astype(np.float64), lens)
    Xp = np.zeros((S * N, B, C_))
    Xp[:T] = Xr
    A = round_bf16(Xp.reshape(S, N, B, C_).transpose(0, 2, 1, 3).reshape(S * B, K))
    want = np.maximum(A @ round_bf16(W.astype(np.float64)).T + bias, 0.0)
    valid = (np.arange(S)[:, None] < (-(-lens // N))[None, :]).reshape(-1)
    want = round_bf16(want * valid[:, None])
    got = host(E)
    np.testing.assert_allclose(got[:, :F], want, rtol=1e-2, atol=1e-2)          # bf16 output: one ulp where fp32 vs fp64 sums straddle
    assert (got[:, F:] == 7.0).all()                                              # padding columns untouched (the ones column lives there)
    assert np.abs(got[:, :F] - want).mean() < 2e-4
    # the packed im2row copy emitted on the way is e2t_conv_pack's, bit for bit (columns >= K untouched)
    Aref = torch.full((S * B, lda), 3.0, dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_conv_pack(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, Aref.data_ptr(), lda, st())
    torch.cuda.synchronize()
    assert torch.equal(Ap[:, :K].view(torch.int16), Aref[:, :K].view(torch.int16))
    assert (host(Ap[:, K:]) == 3.0).all()
    # without the copy: same E
    E2 = torch.full((S * B, lde), 7.0, dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_conv_fwd_fused(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, WT.data_ptr(), ldw, E2.data_ptr(), lde, F, None, 0, C.byref(ep), st())
    torch.cuda.synchronize()
    assert torch.equal(E.view(torch.int16), E2.view(torch.int16))


@pytest.mark.parametrize('M,N,K,nb', [(2049, 3586, 1100, 1), (1024, 4352, 2100, 2)])
def test_gemm_tn_256_tile_instance(hl, M, N, K, nb):
    """The 256 x 256 K-major instance (both output dimensions >= 1024: cfg4's weight gradients), ragged edges, split-K
    slabs + reduction, batched: against NumPy on the bf16-rounded operands."""
    rng = np.random.default_rng(M + N + K)
    lda, ldb = nb * r8(M) + 8, nb * r8(N)
    A = rng.standard_normal((K, lda)); Bm = rng.standard_normal((K, ldb))
    a, b = dev_bf16(A), dev_bf16(Bm)
    wsb = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 1.0
    ep.flags = hl.GEMM_SPLITK
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    if nb > 1:
        ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = nb, r8(M), r8(N), M * r8(N)
    tile, splits = C.c_int(0), C.c_int(0)
    hl.lib.e2t_gemm_plan(1, M, N, K, C.byref(ep), C.byref(tile), C.byref(splits))
    assert tile.value == 256, 'case must exercise the 256 x 256 K-major instance'
    c = torch.full((nb, M, r8(N)), 7.0, dtype=torch.float32, device='cuda')
    hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), r8(N), M, N, K, C.byref(ep), st())
    torch.cuda.synchronize()
    got = host(c)
    for z in range(nb):
        want = round_bf16(A[:, z * r8(M):z * r8(M) + M]).T @ round_bf16(Bm[:, z * r8(N):z * r8(N) + N])
        np.testing.assert_allclose(got[z][:, :N], want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
        assert np.all(got[z][:, N:] == 7.0)


@pytest.mark.parametrize('M,N,K', [(2049, 3841, 1500), (1812, 4353, 1100), (2049, 3846, 1100)])
def test_gemm_tn_256_ragged_edge_leaves_as_its_own_product(hl, M, N, K):
    """A large K-major product whose M (N) ends a few rows (columns) behind a multiple of 256 -- [x | 1]^T . dG with the ones column
    -- is launched as the full tiles + an edge product: same result as one product, incl. the bias-column diversion
    (last_col_out), accumulation onto C and alpha."""
    rng = np.random.default_rng(M * 3 + N)
    lda, ldb = r8(M) + 8, r8(N)
    A = rng.standard_normal((K, lda)); Bm = rng.standard_normal((K, ldb))
    a, b = dev_bf16(A), dev_bf16(Bm)
    wsb = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 0.5
    ep.flags = hl.GEMM_SPLITK | hl.GEMM_ACCUMULATE
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    lc = torch.full((M,), 3.0, dtype=torch.float32, device='cuda')
    ep.last_col_out = lc.data_ptr()
    tile = C.c_int(0)
    hl.lib.e2t_gemm_plan(1, M, N, K, C.byref(ep), C.byref(tile), None)
    assert tile.value == 256
    ldc = N - 1
    c0 = rng.standard_normal((M, ldc)).astype(np.float32)
    c = torch.from_numpy(c0.copy()).cuda()
    hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc, M, N, K, C.byref(ep), st())
    torch.cuda.synchronize()
    want = 0.5 * (round_bf16(A[:, :M]).T @ round_bf16(Bm[:, :N]))
    np.testing.assert_allclose(host(c), c0 + want[:, :N - 1], rtol=1e-5, atol=1e-4 * np.sqrt(K))
    np.testing.assert_allclose(host(lc), want[:, N - 1], rtol=1e-5, atol=1e-4 * np.sqrt(K))


@pytest.mark.parametrize('M,N,K,acc', [(2049, 3841, 1500, False), (2049, 4096, 8704, True), (1793, 4100, 700, False)])
def test_gemm_tn_256_ones_row_is_a_column_sum_pass(hl, M, N, K, acc):
    """E2T_GEMM_LAST_ROW_ONES (ABI 8): [x | 1]^T . dG whose ones column is the ONE row beyond the last full 256-row tile -- the
    bias row of the product is taken by a column-sum pass over B (k_colsum_bf16) instead of an edge product with slabs and a
    reduction kernel of its own.  Same values as the plain product (fp32 sums of bf16 values in another order), alpha and
    accumulation applied, nothing else of C touched; without the flag the call takes the edge product."""
    rng = np.random.default_rng(M + N + K)
    lda, ldb = r8(M) + 8, (N + 63) // 64 * 64
    A = rng.standard_normal((K, lda)); A[:, M - 1] = 1.0
    Bm = rng.standard_normal((K, ldb))
    a, b = dev_bf16(A), dev_bf16(Bm)
    wsb = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device='cuda')
    want = 0.5 * (round_bf16(A[:, :M]).T @ round_bf16(Bm[:, :N]))
    outs = []
    c0 = rng.standard_normal((M + 1, N)).astype(np.float32) if acc else np.full((M + 1, N), 7.0, np.float32)
    for flag in (hl.GEMM_LAST_ROW_ONES, 0):
        ep = hl.GemmEpilogue(); ep.alpha = 0.5
        ep.flags = hl.GEMM_SPLITK | flag | (hl.GEMM_ACCUMULATE if acc else 0)
        ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        tile = C.c_int(0)
        hl.lib.e2t_gemm_plan(1, M, N, K, C.byref(ep), C.byref(tile), None)
        assert tile.value == 256
        c = torch.from_numpy(c0.copy()).cuda()
        hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, M, N, K, C.byref(ep), st())
        torch.cuda.synchronize()
        got = host(c)
        np.testing.assert_allclose(got[:M], (c0[:M] if acc else 0) + want, rtol=1e-5, atol=1e-4 * np.sqrt(K))
        assert np.array_equal(got[M], c0[M])                                   # the row behind the product is nobody's
        outs.append(got)
    assert np.array_equal(outs[0][:M - 1], outs[1][:M - 1])                    # the full tiles are the same launch either way


def test_gemm_nt_last_round_leaves_as_a_second_launch(hl):
    """A large K-contiguous product whose tile count ends a little behind a whole number of rounds of resident workgroups (cfg4's
    input gradient 8704 x 2048: 1088 tiles of 128 x 128 = 2.125 rounds of 512) is launched as the whole rounds + a short second
    launch (K split) over the remaining rows.  Same result as one product -- also for what depends on the ROW INDEX: the row-length
    mask and the dropout counter."""
    rng = np.random.default_rng(5)
    M, N, K, rowsB = 8704, 2048, 2048, 256
    A, Bm = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    a, b = dev_bf16(A), dev_bf16(Bm)
    wsb = torch.zeros(16 * 1024 * 1024, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 0.5
    ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    want = 0.5 * (round_bf16(A).astype(np.float32) @ round_bf16(Bm).astype(np.float32).T)
    if True:        # (the epilogue-rich form: everything that depends on the row index)
        lens = rng.integers(20, M // rowsB + 1, size=rowsB); lens[3] = M // rowsB
        lt = torch.tensor(lens, dtype=torch.int32, device='cuda')
        step = torch.tensor([2], dtype=torch.int32, device='cuda')
        c0 = rng.standard_normal((M, N)).astype(np.float32)
        c = torch.from_numpy(c0.copy()).cuda()
        ep.flags = hl.GEMM_ACCUMULATE | hl.GEMM_DROPOUT
        ep.drop_rate, ep.drop_seed, ep.drop_step, ep.drop_stream, ep.drop_ld = 0.5, 77, step.data_ptr(), 4, N
        ep.row_lens, ep.rows_per_step = lt.data_ptr(), rowsB
        tile = C.c_int(0)
        hl.lib.e2t_gemm_plan(0, M, N, K, C.byref(ep), C.byref(tile), None)
        assert tile.value == 128
        hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N, K, C.byref(ep), st())
        torch.cuda.synchronize()
        valid = (np.arange(M) // rowsB) < lens[np.arange(M) % rowsB]
        ref = c0 + want * (keep_mask((M, N), 0.5, 79, 4) / 0.5) * valid[:, None]
        np.testing.assert_allclose(host(c), ref, rtol=1e-4, atol=2e-2)


def test_gemm_stamp_hook_times_every_launch(hl):
    """e2t_gemm_stamps (bench.py's live in-step timing): with a buffer set, every GEMM launch records [first workgroup start, last
    workgroup end] on the 100-MHz clock in the next slot, tagged with its instance; off again, nothing is written."""
    rng = np.random.default_rng(1)
    M, N, K = 300, 260, 512
    a, b = dev_bf16(rng.standard_normal((M, K))), dev_bf16(rng.standard_normal((N, K)))
    c = torch.zeros(M, N, dtype=torch.float32, device='cuda')
    stamps = torch.tensor([-1, 0] * 8, dtype=torch.int64, device='cuda')
    hl.lib.e2t_gemm_stamps(stamps.data_ptr(), 8)
    try:
        for _ in range(3):
            hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N, K, None, st())
        torch.cuda.synchronize()
        kinds = (C.c_int * 8)()
        hl.lib.e2t_gemm_stamp_kinds(kinds, 8)
    finally:
        hl.lib.e2t_gemm_stamps(None, 0)
    sv = stamps.cpu().numpy().view(np.uint64).reshape(8, 2)
    assert list(kinds)[:3] == [3, 3, 3] and list(kinds)[3:] == [-1] * 5              # E2T_GEMM_KIND_NT128
    for k in range(3):
        us = (int(sv[k, 1]) - int(sv[k, 0])) * 0.01
        assert 0.5 < us < 500.0, us
    assert sv[0, 0] <= sv[1, 0] <= sv[2, 0]                                            # launches of one stream, in order
    assert np.all(sv[3:, 1] == 0)
    before = stamps.clone()
    hl.lib.e2t_gemm_nt_bf16(a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), N, M, N, K, None, st())
    torch.cuda.synchronize()
    assert torch.equal(before, stamps)


@pytest.mark.parametrize('G', [1, 2, 3])
def test_grouped_row_order_of_the_conv_stack(hl, G):
    """e2t_conv_pack_grouped / e2t_conv_unpack_grad_grouped / e2t_gemm_epilogue.row_group: row m = (tg*B + b)*G + g holds step
    t' = tg*G + g (the order in which a stack of strided conv layers keeps the rows of its lower layers, trainers.py:406-407).
    The grouped im2row is the plain one with its rows permuted, the un-im2row inverts it, and the GEMM's row-length mask zeroes
    the rows whose step lies beyond the utterance's decimated length."""
    rng = np.random.default_rng(G)
    B, T, N, C_ = 5, 29, 2, 8
    X = np.abs(rng.standard_normal((B, T, C_))) + 0.1
    lens = np.array([29, 1, 0, 13, 24])
    for b in range(B):
        X[b, lens[b]:] = 0
    xt = torch.tensor(X, dtype=torch.float32, device='cuda')
    lt = torch.tensor(lens, dtype=torch.int32, device='cuda')
    S1 = -(-T // N)                                   # plain: steps of this layer
    S = -(-T // (N * G)) * G                          # grouped: a whole number of groups
    K8 = r8(N * C_)
    plain = torch.zeros((S1 * B, K8), dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_conv_pack(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, plain.data_ptr(), K8, st())
    grp = torch.full((S * B, K8), 3.0, dtype=torch.bfloat16, device='cuda')
    hl.lib.e2t_conv_pack_grouped(xt.data_ptr(), lt.data_ptr(), B, T, C_, N, G, grp.data_ptr(), K8, st())
    torch.cuda.synchronize()
    P_, Gr = host(plain), host(grp)
    m = np.arange(S * B)
    tp, b = (m // (B * G)) * G + m % G, (m // G) % B
    want = np.zeros((S * B, K8))
    ok = tp < S1
    want[ok] = P_[tp[ok] * B + b[ok]]
    np.testing.assert_array_equal(Gr, want)
    # un-im2row of a gradient laid out in the grouped order == the plain one on the permuted rows
    dA = rng.standard_normal((S * B, K8)).astype(np.float32)
    dAp = np.zeros((S1 * B, K8), np.float32)
    dAp[tp[ok] * B + b[ok]] = dA[ok]
    d1 = torch.zeros(B, T, C_, dtype=torch.float32, device='cuda'); d2 = torch.zeros_like(d1)
    a1, a2 = torch.tensor(dAp, device='cuda'), torch.tensor(dA, device='cuda')
    hl.lib.e2t_conv_unpack_grad(a1.data_ptr(), K8, lt.data_ptr(), B, T, C_, N, d1.data_ptr(), st())
    hl.lib.e2t_conv_unpack_grad_grouped(a2.data_ptr(), K8, lt.data_ptr(), B, T, C_, N, G, d2.data_ptr(), st())
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and float(d1.abs().max()) > 0
    # row mask of the product in the grouped order
    F = 16
    W = rng.standard_normal((F, K8))
    wt = dev_bf16(W)
    ld_t = torch.tensor(-(-lens // N), dtype=torch.int32, device='cuda')
    out = torch.full((S * B, F), 7.0, dtype=torch.float32, device='cuda')
    ep = hl.GemmEpilogue(); ep.alpha = 1.0
    ep.row_lens, ep.rows_per_step, ep.row_group = ld_t.data_ptr(), B, G
    hl.lib.e2t_gemm_nt_bf16(grp.data_ptr(), K8, wt.data_ptr(), K8, out.data_ptr(), F, S * B, F, K8, C.byref(ep), st())
    torch.cuda.synchronize()
    ref = Gr @ round_bf16(W).T * (tp < (-(-lens // N))[b])[:, None]
    np.testing.assert_allclose(host(out), ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('V', [1806, 2500, 950])
def test_greedy_step_argmax_and_bookkeeping(hl, V):
    """e2t_greedy_step: arg-max of each row's logits (lowest index on ties, as numpy), token recorded unless the utterance is done,
    <EOS> latches `done`, the token is the next step's input.  Rows that fit the registers (V <= 2048) and longer ones."""
    rng = np.random.default_rng(9)
    B, L, l = 37, 7, 3
    lg = rng.standard_normal((B, V)).astype(np.float32)
    lg[5, 100] = lg[5, 900] = 9.0                      # a tie: the lower index wins
    lg[6, 1] = 50.0                                    # <EOS>
    done0 = (rng.random(B) < 0.3).astype(np.int32)
    done0[6] = 0
    lt = torch.tensor(lg, device='cuda')
    done = torch.tensor(done0, device='cuda')
    out = torch.full((B, L), -5, dtype=torch.int32, device='cuda')
    nxt = torch.full((B,), -5, dtype=torch.int32, device='cuda')
    hl.lib.e2t_greedy_step(lt.data_ptr(), V, B, V, l, L, 1, 0, done.data_ptr(), out.data_ptr(), nxt.data_ptr(), st())
    torch.cuda.synchronize()
    arg = lg.argmax(1)
    assert arg[5] == 100
    np.testing.assert_array_equal(host(nxt), arg)
    np.testing.assert_array_equal(host(out)[:, l], np.where(done0 != 0, 0, arg))
    assert np.all(np.delete(host(out), l, axis=1) == -5)
    np.testing.assert_array_equal(host(done), done0 | (arg == 1))


@pytest.mark.parametrize('R,Cc,s0,update', [(64, 64, 64, True), (100, 132, 132, True), (401, 1600, 1600, True), (130, 72, 200, True),
                                            (37, 12, 12, False), (801, 260, 260, True)])
def test_adam_pack_tiles_equal_the_separate_kernels_bit_for_bit(hl, R, Cc, s0, update):
    """e2t_adam_pack_batch (ABI 8): the optimiser update of a sub-matrix of the flat buffers and ALL its bf16 images in one pass --
    against e2t_adam_ema_step followed by the single-image pack entry points on the same data: masters, Adam state, EMA shadows
    and every image (plain cast, transposed cast, the three fragment layouts) bit for bit; nothing outside the sub-matrix moves
    (ragged edges: R, C not multiples of the 64 x 64 tile; s0 > C: a column block of a wider matrix; update=False: images only)."""
    rng = np.random.default_rng(R + Cc)
    off = 8                                               # the sub-matrix starts inside the buffers
    n = off + R * s0 + 12
    f = lambda scale=1.0: torch.tensor(scale * rng.standard_normal(n), dtype=torch.float32, device='cuda')
    bufs = {k: f() for k in 'pgme'}
    bufs['v'] = f().abs() * 1e-2
    ref = {k: t.clone() for k, t in bufs.items()}
    step = torch.full((1,), 3, dtype=torch.int32, device='cuda')
    h = hl.AdamHyper(1e-2, 0.9, 0.999, 1e-8, 0.99, 0.5, 1, None)
    KBc, KBr = (Cc + 31) // 32, (R + 31) // 32
    NTr, NTc = (R + 15) // 16, (Cc + 15) // 16
    UT = (Cc // 4 + 15) // 16
    ldc, ldr = r8(Cc) + 8, r8(R)
    bf = lambda *shape: torch.full(shape, 7.0, dtype=torch.bfloat16, device='cuda')
    imgs = {1: bf(R, ldc), 2: bf(Cc, ldr), 3: bf(NTr * KBc * 64 * 8), 4: bf(NTc * KBr * 64 * 8), 5: bf(4 * UT * KBr * 64 * 8)}
    want = {k: t.clone() for k, t in imgs.items()}
    # ---- reference: the separate kernels
    if update:
        for r in range(R):                               # (row by row: the range kernel knows nothing of strides)
            o = 4 * (off + r * s0)
            hl.lib.e2t_adam_ema_step(ref['p'].data_ptr() + o, ref['g'].data_ptr() + o, ref['m'].data_ptr() + o, ref['v'].data_ptr() + o,
                                     ref['e'].data_ptr() + o, Cc, step.data_ptr(), C.byref(h), st())
    src = ref['p'].data_ptr() + 4 * off
    hl.lib.e2t_cast_pack(src, s0, 1, R, Cc, want[1].data_ptr(), ldc, st())
    hl.lib.e2t_cast_pack(src, 1, s0, Cc, R, want[2].data_ptr(), ldr, st())
    hl.lib.e2t_pack_frag(src, s0, 1, R, Cc, want[3].data_ptr(), st())
    hl.lib.e2t_pack_frag(src, 1, s0, Cc, R, want[4].data_ptr(), st())
    for g in range(4):
        hl.lib.e2t_pack_frag(src + 4 * g, 4, s0, Cc // 4, R, want[5].data_ptr() + 2 * g * UT * KBr * 512, st())
    # ---- the tile kernel: two descriptors (three images + two images of the same sub-matrix: the second one packs only)
    def table(kinds, first):
        d = hl.TileDesc()
        d.first_block, d.R, d.C, d.nimg, d.src_off, d.s0 = first, R, Cc, len(kinds), off, s0
        for j, k in enumerate(kinds):
            d.img[j].dst, d.img[j].kind, d.img[j].ld = imgs[k].data_ptr(), k, {1: ldc, 2: ldr, 3: KBc, 4: KBr, 5: KBr}[k]
        return d
    nb = ((R + 63) // 64) * ((Cc + 63) // 64)
    for kinds, upd in (([1, 2, 3], update), ([4, 5], False)):
        raw = bytes(table(kinds, 0))
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        hl.lib.e2t_adam_pack_batch(dev.data_ptr(), 1, nb, bufs['p'].data_ptr(), bufs['g'].data_ptr(), bufs['m'].data_ptr(), bufs['v'].data_ptr(),
                                   bufs['e'].data_ptr(), step.data_ptr(), C.byref(h) if upd else None, st())
    torch.cuda.synchronize()
    for k in 'pmve':
        assert torch.equal(bufs[k], ref[k]), k
    for k in imgs:                                        # (padding columns of the cast images keep their fill on both sides)
        assert torch.equal(imgs[k].view(torch.int16), want[k].view(torch.int16)), k


def test_adam_pack_skips_when_the_error_word_is_set(hl):
    rng = np.random.default_rng(0)
    n = 64 * 64
    bufs = [torch.tensor(np.abs(rng.standard_normal(n)), dtype=torch.float32, device='cuda') for _ in range(5)]      # (p, g, m, v, ema; v >= 0)
    keep = [t.clone() for t in bufs]
    img = torch.full((64, 64), 7.0, dtype=torch.bfloat16, device='cuda')
    step = torch.ones(1, dtype=torch.int32, device='cuda')
    skip = torch.ones(1, dtype=torch.int32, device='cuda')
    h = hl.AdamHyper(1e-2, 0.9, 0.999, 1e-8, 0.99, 1.0, 0, skip.data_ptr())
    d = hl.TileDesc()
    d.first_block, d.R, d.C, d.nimg, d.src_off, d.s0 = 0, 64, 64, 1, 0, 64
    d.img[0].dst, d.img[0].kind, d.img[0].ld = img.data_ptr(), 1, 64
    dev = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).cuda()
    args = [dev.data_ptr(), 1, 1] + [t.data_ptr() for t in bufs] + [step.data_ptr(), C.byref(h), st()]
    hl.lib.e2t_adam_pack_batch(*args)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(bufs, keep)) and bool((img.float() == 7.0).all())
    skip.zero_()
    hl.lib.e2t_adam_pack_batch(*args)
    torch.cuda.synchronize()
    assert not torch.equal(bufs[0], keep[0]) and torch.equal(img, bufs[0].view(64, 64).bfloat16())


@pytest.mark.parametrize('M,N,K,nb', [(401, 1600, 8704, 2), (801, 3200, 8704, 1), (1024, 4096, 8704, 2)])
def test_gemm_tn_keep_slabs_sum_to_the_reduced_product(hl, M, N, K, nb):
    """E2T_GEMM_KEEP_SLABS (ABI 8): a split K-major product whose reduction would be plain leaves its slabs and says where
    (e2t_slab_info); their sum in split order is, bit for bit, what the same call writes to C without the flag.  A product that
    does not qualify (here: alpha != 1) is reduced as ever and reports splits = 1."""
    rng = np.random.default_rng(M + N)
    lda, ldb = nb * r8(M) + 8, nb * r8(N)
    a, b = dev_bf16(rng.standard_normal((K, lda))), dev_bf16(rng.standard_normal((K, ldb)))
    wsb = torch.zeros(48 * 1024 * 1024, dtype=torch.float32, device='cuda')

    def call(flags, alpha=1.0):
        ep = hl.GemmEpilogue(); ep.alpha = alpha
        ep.flags = hl.GEMM_SPLITK | flags
        ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        if nb > 1:
            ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = nb, r8(M), r8(N), M * N
        info = hl.SlabInfo()
        ep.slabs_out = C.pointer(info)
        c = torch.full((nb, M, N), 7.0, dtype=torch.float32, device='cuda')
        hl.lib.e2t_gemm_tn_bf16(a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, M, N, K, C.byref(ep), st())
        torch.cuda.synchronize()
        return c, info
    want, info0 = call(0)
    assert info0.splits == 1 and not info0.slab
    c, info = call(hl.GEMM_KEEP_SLABS)
    assert info.splits >= 2 and info.batch == nb and info.stride == M * N, (info.splits, info.batch, info.stride)
    assert bool((c == 7.0).all())                                             # C was not touched
    off = (info.slab - wsb.data_ptr()) // 4
    slabs = wsb[off:off + nb * info.splits * M * N].view(nb, info.splits, M, N)
    acc = slabs[:, 0].clone()
    for s in range(1, info.splits):
        acc += slabs[:, s]
    assert torch.equal(acc, want)
    c2, info2 = call(hl.GEMM_KEEP_SLABS, alpha=0.5)
    assert info2.splits == 1 and torch.equal(c2, 0.5 * want)


@pytest.mark.parametrize('B,V,K', [(1, 1806, 800), (3, 1806, 800), (8, 950, 2048), (2, 70, 52), (5, 2500, 136)])
def test_greedy_head_small_equals_projection_argmax_and_gather(hl, B, V, K):
    """e2t_decode_init + e2t_greedy_head_small (ABI 9; the online predictor's one utterance per call, trainers.py:925-949): logits =
    bias + WT . h in fp32 from bf16 operands, arg-max (lowest index on ties), the bookkeeping of e2t_greedy_step, and the chosen
    tokens' rows of the input-projection table copied for the next step -- one launch.  Against numpy on the rounded operands;
    launched three times in a row on one scratch buffer (the ticket word is left at zero by the last workgroup to arrive)."""
    rng = np.random.default_rng(B * 1000 + V + K)
    L, eos, pad = 6, 1, 0
    ldh, ldw = r8(K) + 16, r8(K) + 8
    Hn, Wn = rng.standard_normal((B, K)), rng.standard_normal((V, K)) * 0.2
    bias = rng.standard_normal(V).astype(np.float32)
    hbuf = torch.zeros(B, ldh, dtype=torch.bfloat16, device='cuda'); hbuf[:, :K] = dev_bf16(Hn)
    hbuf[:, K:] = 1.0                                              # (the ones column / padding beyond K must not enter the sums)
    wbuf = torch.zeros(V, ldw, dtype=torch.bfloat16, device='cuda'); wbuf[:, :K] = dev_bf16(Wn)
    bt = torch.tensor(bias, device='cuda')
    rw = 24                                                        # 32-bit words per table row
    table = torch.arange(V * rw, dtype=torch.int32, device='cuda').reshape(V, rw).contiguous()
    done, hyp = torch.full((B,), 7, dtype=torch.int32, device='cuda'), torch.full((B, L), -3, dtype=torch.int32, device='cuda')
    tok0, dlens = torch.full((B,), -3, dtype=torch.int32, device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda')
    hl.lib.e2t_decode_init(done.data_ptr(), hyp.data_ptr(), tok0.data_ptr(), dlens.data_ptr(), B, L, eos, pad, st())
    torch.cuda.synchronize()
    assert not host(done).any() and (host(hyp) == pad).all() and (host(tok0) == eos).all() and (host(dlens) == L).all()
    logits = round_bf16(Hn) @ round_bf16(Wn).T + bias
    want = (round_bf16(Hn) @ round_bf16(Wn).T + bias).argmax(1)
    top2 = np.sort(round_bf16(Hn) @ round_bf16(Wn).T + bias, 1)[:, -2:]
    assert ((top2[:, 1] - top2[:, 0]) > 1e-4).all()               # (random logits: the decision is clear of fp32 summation order)
    scratch = torch.zeros(2 + 2 * 64 * 8, dtype=torch.int32, device='cuda')
    nxt, gx = torch.full((B,), -3, dtype=torch.int32, device='cuda'), torch.full((B, rw), -1, dtype=torch.int32, device='cuda')
    done0 = np.zeros(B, np.int32)
    if B > 2:
        done[2] = 1; done0[2] = 1
    for l in range(3):                                             # same inputs three times: the ticket word resets itself
        hl.lib.e2t_greedy_head_small(hbuf.data_ptr(), ldh, wbuf.data_ptr(), ldw, bt.data_ptr(), B, V, K, l, L, eos, pad, done.data_ptr(),
                                     hyp.data_ptr(), nxt.data_ptr(), table.data_ptr(), rw, gx.data_ptr(), scratch.data_ptr(), st())
        torch.cuda.synchronize()
        assert int(scratch[0].item()) == 0
        np.testing.assert_array_equal(host(nxt), want)
        np.testing.assert_array_equal(host(hyp)[:, l], np.where(done0 != 0, pad, want))
        np.testing.assert_array_equal(host(gx), host(table)[want])
        done0 = done0 | (want == eos)
        np.testing.assert_array_equal(host(done), done0)
    assert (host(hyp)[:, 3:] == pad).all()
    # ties: two identical vocabulary rows -> the lower index; no bias, no table
    wbuf[V - 1] = wbuf[3]; wbuf[3] *= 0; wbuf[V - 1] *= 0
    wbuf[10, :K] = 4.0 * hbuf[0, :K]; wbuf[V - 2, :K] = 4.0 * hbuf[0, :K]
    hl.lib.e2t_greedy_head_small(hbuf.data_ptr(), ldh, wbuf.data_ptr(), ldw, None, 1, V, K, 4, L, eos, pad, done.data_ptr(),
                                 hyp.data_ptr(), nxt.data_ptr(), None, 0, None, scratch.data_ptr(), st())
    torch.cuda.synchronize()
    assert int(nxt[0].item()) == 10
