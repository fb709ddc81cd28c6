"""bfloat16 rounding emulation (round-to-nearest-even), test infrastructure.

Used by oracle/seq2seq.py to round tensors at exactly the points where the HIP
path stores an MFMA operand in bf16 ("Rounding points").
"""
import numpy as np


def round_bf16(x):
    """Round an array to the nearest bfloat16 (ties to even); returns float64."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = (u + np.uint64(0x7FFF) + lsb) & np.uint64(0xFFFF0000)
    out = r.astype(np.uint32).view(np.float32)
    # NaN/Inf pass through untouched
    bad = ~np.isfinite(a)
    if bad.any():
        out = np.where(bad, a, out)
    return out.astype(np.float64).reshape(np.shape(x))


def to_bf16_bits(x):
    """uint16 bit patterns of round_bf16(x) (for feeding device buffers)."""
    a = np.ascontiguousarray(round_bf16(x).astype(np.float32))
    return (a.view(np.uint32) >> np.uint32(16)).astype(np.uint16).reshape(np.shape(x))


def from_bf16_bits(bits):
    b = np.ascontiguousarray(np.asarray(bits, dtype=np.uint16))
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32).astype(np.float64)
