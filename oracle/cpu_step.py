"""ctypes front of oracle/cpu_step.cpp -- the C++17 / OpenMP fp32 train step that SURVEY.md 8 d5 (i) asks for as the CPU number
timed beside the GPU run.  TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py): used by tests/test_cpu_step.py (pinned
against the NumPy oracle) and by bench.py's `cpu_baseline`; the product path never imports it."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import seq2seq as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_cpu', 'libe2t_cpu_step.so')


def _host():
    try:
        with open('/proc/cpuinfo') as f:
            return next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), 'unknown')
    except OSError:
        return 'unknown'


def _usable_cpus():
    """Hardware threads this process may use: the affinity mask, capped by the cgroup's CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def build(force=False):
    """g++ -O3 -march=native -fopenmp.  The library is built FOR the box that runs it: a copy built on another CPU model (the
    in-tree file travels with the repository snapshot) is rebuilt."""
    src = os.path.join(HERE, 'cpu_step.cpp')
    tag = SO + '.host'
    host = _host()
    built_for = open(tag).read() if os.path.exists(tag) else None
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src) or built_for != host:
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.run(['g++', '-O3', '-march=native', '-fopenmp', '-std=c++17', '-shared', '-fPIC', src, '-o', SO], check=True)
        with open(tag, 'w') as f:
            f.write(host)
    return SO


def _names(spec, sid):
    """Oracle parameter names in the order of the flat array (cpu_step.cpp: e2t_cpu_create)."""
    (cn, ci, co, n), = O.conv_layers(spec, sid)
    out = [cn + '/weights', cn + '/biases']
    for l in range(len(spec.enc_rnn)):
        for d in ('fw', 'bw'):
            out += ['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, d), 'seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, d)]
    if spec.aux_layer is not None:
        sizes = [2 * spec.enc_rnn[spec.aux_layer]] + list(spec.aux_hidden) + [spec.aux_dim]
        for nm in O.ff_names('encoder_%d_projection' % spec.aux_layer, sizes):
            out += [nm + '/weights', nm + '/biases']
    out.append('seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed))
    out += ['seq2seq/decoder_rnn/cell_0/kernel', 'seq2seq/decoder_rnn/cell_0/bias']
    for nm in O.ff_names('decoder_projection', [spec.dec_rnn, spec.vocab]):
        out += [nm + '/weights', nm + '/biases']
    return out


class CpuStep:
    def __init__(self, spec, sid, B, T, L, threads=None):
        assert not spec.conv_pre and not spec.aux_extra and not spec.dec_proj_hidden and spec.conv_relu and spec.aux_dist == 'Gaussian'
        self.lib = lib = C.CDLL(build())
        lib.e2t_cpu_create.restype = C.c_void_p
        lib.e2t_cpu_create.argtypes = [C.c_void_p, C.c_void_p]
        for f in ('e2t_cpu_params', 'e2t_cpu_grads', 'e2t_cpu_ema'):
            getattr(lib, f).restype = C.POINTER(C.c_float)
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.e2t_cpu_num_params.restype = C.c_long
        lib.e2t_cpu_num_params.argtypes = [C.c_void_p]
        lib.e2t_cpu_fwd_bwd.argtypes = [C.c_void_p] * 6 + [C.c_int]
        lib.e2t_cpu_adam.argtypes = [C.c_void_p] + [C.c_float] * 5
        lib.e2t_cpu_losses.argtypes = [C.c_void_p, C.c_void_p]
        lib.e2t_cpu_init_ema.argtypes = [C.c_void_p]
        lib.e2t_cpu_destroy.argtypes = [C.c_void_p]
        if threads:
            lib.e2t_cpu_set_threads(int(threads))
        self.spec, self.sid, self.B, self.T, self.L = spec, sid, B, T, L
        aux = spec.aux_layer if spec.aux_layer is not None else -1
        cfg = [spec.channels[sid], spec.decimation, spec.enc_embed, len(spec.enc_rnn)] + list(spec.enc_rnn) + \
              [spec.dec_embed, spec.dec_rnn, spec.vocab, aux, len(spec.aux_hidden)] + list(spec.aux_hidden) + [spec.aux_dim, B, T, L]
        icfg = np.asarray(cfg, np.int32)
        fcfg = np.asarray([spec.ff_dropout, spec.rnn_dropout, spec.forget_bias, spec.aux_scale, spec.dec_scale], np.float32)
        self.h = C.c_void_p(lib.e2t_cpu_create(icfg.ctypes.data, fcfg.ctypes.data))
        self.n = lib.e2t_cpu_num_params(self.h)
        self.names = _names(spec, sid)

    def _view(self, fn):
        return np.ctypeslib.as_array(getattr(self.lib, fn)(self.h), shape=(self.n,))

    def load_params(self, P):
        flat = np.concatenate([np.asarray(P[k], np.float32).reshape(-1) for k in self.names])
        assert flat.size == self.n, (flat.size, self.n)
        self._view('e2t_cpu_params')[:] = flat
        self.lib.e2t_cpu_init_ema(self.h)
        self.shapes = [np.asarray(P[k]).shape for k in self.names]

    def _unflat(self, flat):
        out, o = {}, 0
        for k, sh in zip(self.names, self.shapes):
            n = int(np.prod(sh))
            out[k] = flat[o:o + n].reshape(sh).copy()
            o += n
        return out

    def params(self):
        return self._unflat(self._view('e2t_cpu_params'))

    def grads(self):
        return self._unflat(self._view('e2t_cpu_grads'))

    def fwd_bwd(self, batch, train=True):
        if getattr(self, '_keep_id', None) != id(batch):             # (lengths of a batch are derived once: a timed loop re-uses its batch)
            X = np.ascontiguousarray(batch['encoder_inputs'], np.float32)
            Y = np.ascontiguousarray(batch['decoder_targets'], np.int32)
            A = batch.get('encoder_targets')
            lens = O.sequence_lengths(X).astype(np.int32)
            alens = None
            if A is not None:
                A = np.ascontiguousarray(A, np.float32)
                alens = O.sequence_lengths(A).astype(np.int32)
            self._keep, self._keep_id = (X, Y, A, lens, alens), id(batch)
        X, Y, A, lens, alens = self._keep
        self.lib.e2t_cpu_fwd_bwd(self.h, X.ctypes.data, lens.ctypes.data, Y.ctypes.data, A.ctypes.data if A is not None else None,
                                 alens.ctypes.data if A is not None else None, int(train))
        out = (C.c_double * 3)()
        self.lib.e2t_cpu_losses(self.h, out)
        return dict(decoder=out[0], aux=out[1], accuracy=out[2])

    def adam(self, lr=5e-4, b1=0.9, b2=0.999, eps=1e-8, ema_decay=0.99):
        self.lib.e2t_cpu_adam(self.h, lr, b1, b2, eps, ema_decay)

    def step(self, batch):
        losses = self.fwd_bwd(batch, train=True)
        self.adam()
        return losses

    def close(self):
        if self.h:
            self.lib.e2t_cpu_destroy(self.h)
            self.h = None


def bench_main(argv):
    """python -m oracle.cpu_step <cfg> [timed steps]: the timed CPU train step of bench.py's `cpu_baseline` (run as a process of its
    own: its OpenMP pool does not share the box with torch's).  Prints one JSON line."""
    import json
    import sys
    import time
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    cfg = argv[0] if argv else 'cfg2'
    timed = int(argv[1]) if len(argv) > 1 else 5
    kw, B, T, L = bench.CONFIGS[cfg]
    sid = list(kw['channels'])[0]
    kw1 = dict(kw, channels={sid: kw['channels'][sid]})
    spec = O.NetSpec(**kw1)
    batch = bench.synth_batch(kw1, B, T, L, seed=1)
    cpu = CpuStep(spec, sid, B, T, L)
    cpu.load_params(O.init_params(spec, seed=0, dtype=np.float32))
    ncpu = _usable_cpus()
    cpu.lib.e2t_cpu_set_threads(min(ncpu, 16))
    cpu.step(batch)                                            # warm-up: first touch of every buffer
    # ascending thread counts, stopping at the first one that is clearly slower than the best so far: a box whose cgroup gives
    # the process fewer cores than it lists makes every count beyond them catastrophically slow (spinning barriers on shared cores:
    # 256 threads on the GPU box took 197 s per step where 64 took 1.0)
    sweep, nt = [], 8
    while True:
        nt = min(nt, ncpu)
        cpu.lib.e2t_cpu_set_threads(nt)
        t0 = time.perf_counter()
        cpu.step(batch)
        sweep.append((time.perf_counter() - t0, nt))
        if nt >= ncpu or sweep[-1][0] > 1.25 * min(sweep)[0]:
            break
        nt *= 2
    threads = min(sweep)[1]
    cpu.lib.e2t_cpu_set_threads(threads)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        losses = cpu.step(batch)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    print(json.dumps(dict(value=round(B / med, 3), cores=int(threads), s_per_step=med, min=min(ts), max=max(ts),
                          sweep=[[n, round(t, 4)] for t, n in sweep], nproc=ncpu, listed=os.cpu_count(), loss=losses['decoder'], B=B, T=T)))


if __name__ == '__main__':
    import sys
    bench_main(sys.argv[1:])
