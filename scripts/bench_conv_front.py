"""Front-end in isolation: lengths + (im2row pack + GEMM) vs the fused one-pass kernel, at a config's sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts')) if 'ROOT' in globals() else sys.path.insert(0, 'scripts')
import _dbg  # noqa: F401  (debug build of the library: the E2T_* kernel switches and phase stamps live there)
import torch
import bench
from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, capture
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
kw, B, T, L = bench.CONFIGS[cfg]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for fused in ('0', '1', '1a'):
    eng = Seq2SeqEngine(NetSpec(**kw), seed=1, options={'fused_conv': fused[0]})
    eng.init_params(0)
    sid = list(kw['channels'])[0]
    ws = eng.workspace(sid, B, T, L)
    eng.set_batch(ws, bench.synth_batch(kw, B, T, L, 1))
    import ctypes as C
    from ecog2txt_amd import hip_lib as H
    from ecog2txt_amd.hip_lib import lib
    sp = eng.spec
    Cc, N, S, M = ws['C'], sp.decimation, ws['S'], ws['M']
    def front():
        st = eng.stream
        lib.e2t_seq_lengths_tail_f32(ws['X'].data_ptr(), B, T, Cc, N, ws['lens'].data_ptr(), ws['lens_d'].data_ptr(), st)
        if fused != '0':
            ep = H.GemmEpilogue()
            ep.bias = eng.store.ptr('conv%s.W' % sid, eng.store.p, ws['Kc'] * sp.enc_embed)
            ep.alpha, ep.flags = 1.0, H.GEMM_RELU | H.GEMM_OUT_BF16 | H.GEMM_DROPOUT
            ep.row_lens, ep.rows_per_step = ws['lens_d'].data_ptr(), B
            ep.splitk_ws, ep.splitk_ws_bytes = eng.splitk_ws.data_ptr(), eng.splitk_ws.numel() * 4
            ep.drop_rate, ep.drop_stream, ep.drop_ld, ep.drop_seed, ep.drop_step = sp.ff_dropout, 1, sp.enc_embed, eng.seed, eng.step_t.data_ptr()
            lib.e2t_conv_fwd_fused(ws['X'].data_ptr(), ws['lens'].data_ptr(), B, T, Cc, N, eng.convT[sid].data_ptr(), ws['Kc8'],
                                   ws['E'].data_ptr(), eng.F8, sp.enc_embed,
                                   ws['A'].data_ptr() if fused == '1a' else None, ws['Kc8'], C.byref(ep), st)
        else:
            lib.e2t_conv_pack(ws['X'].data_ptr(), ws['lens'].data_ptr(), B, T, Cc, N, ws['A'].data_ptr(), ws['Kc8'], st)
            eng.gemm(ws['A'].data_ptr(), ws['Kc8'], eng.convT[sid].data_ptr(), ws['Kc8'], ws['E'].data_ptr(), eng.F8, M, sp.enc_embed,
                     ws['Kc8'], bias=eng.store.ptr('conv%s.W' % sid, eng.store.p, ws['Kc'] * sp.enc_embed), relu=True, out_bf16=True,
                     drop=(sp.ff_dropout, 1, sp.enc_embed), row_lens=(ws['lens_d'].data_ptr(), B))
    try:
        us = timeit(front)
    except Exception as e:
        us = float('nan'); print(e)
    gb = B * T * kw['channels'][sid] * 4 / 1e9
    print('%s fused=%s: front-end (lengths + conv [+ final_state]) %.1f us; input %.2f GB -> %.2f TB/s of algorithmic input' % (cfg, fused, us, gb, gb / us * 1e3))
    del eng
