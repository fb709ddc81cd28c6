"""What the input-projection GEMM (8704 x 3200 x K, 256 x 256 instance) spends outside its main loop: run per variant in a
process of its own (the debug library reads E2T_GEMM_DBG / E2T_GEMM_TILE once).  usage: probe_gx_gemm.py [child]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import ctypes as C
    import _dbg  # noqa: F401
    import torch
    from ecog2txt_amd import hip_lib as H
    from ecog2txt_amd.hip_lib import lib
    H.load()
    def r8(x): return (x + 7) // 8 * 8
    for M, N, K, obf in [(8704, 3200, 64, 0), (8704, 3200, 64, 1), (8704, 3200, 808, 0), (8704, 3200, 808, 1), (8704, 8192, 2112, 0), (8704, 8192, 2112, 1)]:
        a = torch.randn(M, r8(K), device='cuda').to(torch.bfloat16)
        b = torch.randn(N, r8(K), device='cuda').to(torch.bfloat16)
        c = torch.zeros(M, r8(N), device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
        ep = H.GemmEpilogue(); ep.alpha = 1.0; ep.flags = H.GEMM_OUT_BF16 if obf else 0
        def run():
            lib.e2t_gemm_nt_bf16(a.data_ptr(), r8(K), b.data_ptr(), r8(K), c.data_ptr(), r8(N), M, N, r8(K), C.byref(ep), torch.cuda.current_stream().cuda_stream)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print('   M%5d N%5d K%5d out %s  %7.1f us  %7.1f TFLOP/s' % (M, N, K, 'bf16' if obf else 'fp32', us, 2.0 * M * N * K / us / 1e6))
    sys.exit(0)
for tag, env in [('256 x 256 as shipped', {}), ('256 x 256 without the stores', {'E2T_GEMM_DBG': '4'}), ('256 x 256 loads + epilogue only', {'E2T_GEMM_DBG': '2'}),
                 ('128 x 128', {'E2T_GEMM_TILE': '128'})]:
    print(tag, flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=e)
