"""Philox4x32-10 counter-based RNG in NumPy (test infrastructure).

The HIP kernels draw dropout masks from the same generator
(ecog2txt_amd/csrc/philox.h), keyed by (seed, stream) and indexed by the
logical element index of the time-major tensor being dropped, so the oracle
reproduces every mask bit-for-bit.  Dropout rates come from the manifest keys
FF_dropout / RNN_dropout (reference: mocha-1_word_sequence.yaml:6,13; passed
as 0.0 at assessment, trainers.py:816,822).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32; all inputs uint32 arrays (broadcastable)."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.asarray(c1, dtype=np.uint32) + np.zeros_like(c0)
    c2 = np.asarray(c2, dtype=np.uint32) + np.zeros_like(c0)
    c3 = np.asarray(c3, dtype=np.uint32) + np.zeros_like(c0)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform_u24(n, seed, stream):
    """n uniform 24-bit integers; element e uses counter (e>>2, stream) lane e&3.

    key = (seed & 0xffffffff, seed >> 32).
    """
    e = np.arange(n, dtype=np.uint64)
    ctr = (e >> np.uint64(2))
    c0 = (ctr & _MASK).astype(np.uint32)
    c1 = (ctr >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(n, stream, dtype=np.uint32)
    c3 = np.zeros(n, dtype=np.uint32)
    r = philox4x32(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    lane = (e & np.uint64(3)).astype(np.int64)
    out = np.choose(lane, r)
    return (out >> np.uint32(8)).astype(np.int64)


def keep_mask(shape, rate, seed, stream):
    """Boolean keep-mask with P(keep) = 1 - rate, indexed by C-order element."""
    n = int(np.prod(shape))
    if rate <= 0.0:
        return np.ones(shape, dtype=bool)
    thresh = int(rate * 16777216.0)          # same integer threshold on device
    return (uniform_u24(n, seed, stream) >= thresh).reshape(shape)
