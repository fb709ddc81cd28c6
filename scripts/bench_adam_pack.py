"""Isolated timing of the fused optimiser update + re-pack (e2t_adam_pack_batch) against e2t_adam_ema_step + e2t_pack_batch on one
weight matrix: GB/s over the bytes each form moves.  usage: bench_adam_pack.py [R C]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecog2txt_amd import hip_lib as H
from ecog2txt_amd.hip_lib import lib
H.load()
R, Cc = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 8192)
n = R * Cc
dev = 'cuda'
bufs = [torch.randn(n, device=dev) for _ in range(5)]
bufs[3].abs_()
p, g, m, v, e = bufs
step = torch.ones(1, dtype=torch.int32, device=dev)
h = H.AdamHyper(1e-4, 0.9, 0.999, 1e-8, 0.99, 1.0, 0, None)
st = lambda: torch.cuda.current_stream().cuda_stream
KBr = (R + 31) // 32
img = {1: torch.zeros(R, Cc, dtype=torch.bfloat16, device=dev), 2: torch.zeros(Cc, R, dtype=torch.bfloat16, device=dev),
       3: torch.zeros(R * ((Cc + 31) // 32 * 32), dtype=torch.bfloat16, device=dev), 4: torch.zeros(Cc * KBr * 32, dtype=torch.bfloat16, device=dev),
       5: torch.zeros(Cc * KBr * 32, dtype=torch.bfloat16, device=dev)}
ld = {1: Cc, 2: R, 3: (Cc + 31) // 32, 4: KBr, 5: KBr}


def tile_table(kinds):
    d = H.TileDesc()
    d.first_block, d.R, d.C, d.nimg, d.src_off, d.s0 = 0, R, Cc, len(kinds), 0, Cc
    for j, k in enumerate(kinds):
        d.img[j].dst, d.img[j].kind, d.img[j].ld = img[k].data_ptr(), k, ld[k]
    return torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


nb = ((R + 63) // 64) * ((Cc + 63) // 64)
us = timeit(lambda: lib.e2t_adam_ema_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), n, step.data_ptr(), C.byref(h), st()))
print('%d x %d: k_adam_ema           %7.1f us  %5.2f TB/s (36 B/param)' % (R, Cc, us, 36 * n / us / 1e6))
for kinds, upd in (([], True), ([1], True), ([2], True), ([1, 2], True), ([5, 3], True), ([1, 2], False), ([5, 3], False), ([5, 3, 4], True)):
    t = tile_table(kinds)
    fn = lambda: lib.e2t_adam_pack_batch(t.data_ptr(), 1, nb, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), step.data_ptr(),
                                         C.byref(h) if upd else None, st())
    us = timeit(fn)
    byts = (36 if upd else 4) + 2 * len(kinds)
    print('  tile kernel %-9s images %-10s %7.1f us  %5.2f TB/s (%d B/param)' % ('update +' if upd else 'pack only', kinds, us, byts * n / us / 1e6, byts))
