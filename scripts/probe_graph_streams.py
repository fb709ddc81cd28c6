"""hipGraphLaunch picks the streams of a graph's parallel branches from a pool without a bounds check (seen as a SIGSEGV in
libamdhip64 at the first replay of a two-branch graph: DESIGN.md 5.00, profiles/r03_hip_graph_launch_probe.txt).  This probe looks for the history that triggers it:
  child <pre> <nA> <extra> <nB> [launch_on_side]
    pre    : branches of a warm-up graph captured + launched first (0 = none)
    nA, nB : parallel branches of graph A and of graph B (B is made after `extra` unrelated streams were created)
"""
import os, subprocess, sys
import ctypes
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    pre, nA, extra, nB = (int(x) for x in sys.argv[2:6])
    side_launch = len(sys.argv) > 6
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    x = torch.zeros(64, 1024, device='cuda')
    pool = [torch.cuda.Stream() for _ in range(8)]

    def make(n):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(cur)
            joins = []
            for i in range(n - 1):
                s = pool[i]
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    x[i + 1] += 1
                    j = torch.cuda.Event(); j.record(s); joins.append(j)
            x[0] += 1
            for j in joins:
                cur.wait_event(j)
        return g
    if pre:
        make(pre).replay(); torch.cuda.synchronize()
    a = make(nA); a.replay(); torch.cuda.synchronize()
    keep = []
    for _ in range(extra):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(h)) == 0
        keep.append(h)
    b = make(nB)
    if side_launch:
        with torch.cuda.stream(pool[7]):
            b.replay()
    else:
        b.replay()
    torch.cuda.synchronize()
    print('ok', flush=True)
    sys.exit(0)
# Full matrix (pre x nA x nB x extra x launch stream = 192 processes, ~17 min): `probe_graph_streams.py full`.  Measured on ROCm 7.0.2 /
# torch 2.10: SIGSEGV in 4 of 96 launches from the default stream -- (pre 8, A 2, B 4), (8, 2, 5), (8, 3, 3), (8, 3, 5), no extra
# streams -- and in 0 of 96 launches from a side stream.  Default: those four cases, both ways.
full = len(sys.argv) > 1 and sys.argv[1] == 'full'
cases = [(pre, nA, nB, extra) for pre in (0, 8) for nA in (2, 3) for nB in (2, 3, 4, 5) for extra in range(6)] if full else \
        [(8, 2, 4, 0), (8, 2, 5, 0), (8, 3, 3, 0), (8, 3, 5, 0)]
for pre, nA, nB, extra in cases:
    res = []
    for side in ((), ('side',)):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(pre), str(nA), str(extra), str(nB), *side], capture_output=True, text=True)
        res.append('ok' if r.returncode == 0 else 'rc%d' % r.returncode)
    print('warm-up graph %d branches, A %d, %d unrelated streams, B %d: launch from the default stream %s, from a side stream %s' % (pre, nA, extra, nB, res[0], res[1]), flush=True)
