"""Parameter layout of the ECoG->text network on the device: the architecture description (NetSpec), the flat fp32 buffers
(master / gradient / Adam m / Adam v / EMA shadow) with their segments in BACKWARD order (ParamStore), conversion to and from the
reference's variable grammar (MultiSubjectTrainer.recover_model_sizes, ecog2txt/trainers.py:444-554), and the small layout
helpers every other module of the engine shares (leading-dimension rounding, graph capture).  Host-side plumbing only."""
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional
import contextlib
import ctypes as C
import gc
import os
import re

import numpy as np
import torch

from . import hip_lib as H
from .hip_lib import lib

PAD_ID, EOS_ID, OOV_ID = 0, 1, 2          # trainers.py:191-196

STREAM_CONV, STREAM_ENC, STREAM_DEC_EMB, STREAM_DEC_OUT, STREAM_AUX = 1, 10, 20, 21, 30
STREAM_CONV_PRE = 40      # + index of a conv layer in front of the one that feeds the encoder


@contextlib.contextmanager
def capture(graph):
    """torch.cuda.graph(graph) with the cyclic garbage collector out of the way: a collection that runs DURING a stream
    capture may destroy device objects of unrelated, dead Python objects (another engine's CUDAGraphs, events) -- HIP
    refuses that while capturing and the destructor aborts the process.  (Seen: an exception's traceback kept a dead
    engine alive in a reference cycle until the next capture.)"""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()


def rk(x):
    """Leading dimension of anything that serves as a GEMM K dimension: a multiple of the 64-wide K tile, zero
    padded, so that k_gemm_nt never takes its register-staged K-tail path (measured: 98 -> 75 us on the encoder
    input projection, scripts/bench_gemm_variants.py)."""
    return (x + 63) // 64 * 64


def r8(x):
    return (x + 7) // 8 * 8


def ceil_div(a, b):
    return -(-a // b)


@dataclass
class NetSpec:
    """Network sizes; field-for-field the manifest's layer_sizes & co.
    (mocha-1_word_sequence.yaml:5-14, 56-69)."""
    channels: Dict[object, int]
    decimation: int = 12
    enc_embed: int = 100
    enc_rnn: List[int] = field(default_factory=lambda: [400, 400, 400])
    dec_embed: int = 150
    dec_rnn: int = 800
    dec_proj_hidden: List[int] = field(default_factory=list)
    vocab: int = 1806
    aux_layer: Optional[int] = 1
    aux_hidden: List[int] = field(default_factory=lambda: [225])
    aux_dim: int = 13
    aux_dist: str = 'Gaussian'
    aux_scale: float = 1.0
    # further auxiliary heads (one per additional 'encoder_<k>_targets' data key, trainers.py:94-102): dicts with layer,
    # hidden, dim, dist, scale -- a simple path on the main stream; the first head keeps the overlapped schedule
    aux_extra: List[dict] = field(default_factory=list)
    dec_scale: float = 1.0
    ff_dropout: float = 0.1
    rnn_dropout: float = 0.5
    forget_bias: float = 1.0
    conv_relu: bool = True
    # conv layers in front of the one that feeds the encoder (dicts with out, stride; the strides of the whole stack
    # multiply to `decimation`, trainers.py:406-407; [BUILD-DEFINES] the split is given explicitly -- oracle/seq2seq.py)
    conv_pre: List[dict] = field(default_factory=list)

    def as_dict(self):
        return asdict(self)


def conv_stack(spec, Cc):
    """[(in width, out width, stride)] of a subject's temporal-convolution stack, bottom up (oracle.conv_layers)."""
    outs = [int(p['out']) for p in spec.conv_pre] + [spec.enc_embed]
    strides = [int(p['stride']) for p in spec.conv_pre]
    last = spec.decimation // int(np.prod(strides)) if strides else spec.decimation
    assert last >= 1 and last * int(np.prod(strides or [1])) == spec.decimation, 'the conv strides must multiply to the decimation factor'
    return list(zip([Cc] + outs[:-1], outs, strides + [last]))


def conv_seg(sid, j):
    """Parameter segment of conv layer j of subject sid ([stride*in + 1][out], bias last): the bottom layer keeps the
    single-layer name."""
    return 'conv%s.W' % sid if j == 0 else 'conv%s.W%d' % (sid, j)


def conv_tf_name(sid, j, ci, co):
    return 'seq2seq/subnet_%s/encoder_embedding_%d_%d_%d' % (sid, ci, co, j)


def _tf2int(w, Hh):
    """TF gate-major columns [..., 4H] (i|j|f|o) -> unit-major interleaved (u*4+g)."""
    return w.reshape(w.shape[:-1] + (4, Hh)).swapaxes(-1, -2).reshape(w.shape)


def _int2tf(w, Hh):
    return w.reshape(w.shape[:-1] + (Hh, 4)).swapaxes(-1, -2).reshape(w.shape)


SEG_ALIGN = 64      # floats.  A segment's rows are read and written in 64-column tiles by the fused update (e2t_adam_pack_batch): with the
                    # segment on a 256-B boundary -- and row lengths that are multiples of 32 floats, as every weight matrix of the
                    # configurations has -- a tile row is whole 128-B lines in all five flat buffers; at the former 32-B alignment
                    # every tile row straddled a third line and the four written buffers paid partial-line writes (4.1 instead of
                    # 6.5 TB/s)


def seg_pad(n):
    return (n + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN


class ParamStore:
    """Flat fp32 master / grad / Adam / EMA buffers with named segments.

    Segment order = order in which backward produces the gradients (vocab
    projection first, per-subject conv last) so that contiguous ranges are the
    all-reduce buckets of the data-parallel path."""

    def __init__(self, spec, device):
        self.spec, self.device = spec, device
        self.segs = {}
        self.order = []
        off = 0

        def add(name, *shape):
            nonlocal off
            n = int(np.prod(shape))
            self.segs[name] = (off, tuple(shape))
            self.order.append(name)
            off += seg_pad(n)                  # every segment starts on a 256-B boundary (SEG_ALIGN)

        # decoder projection stack (last layer stored transposed, trainers.py:513-520)
        sizes = [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab]
        for i in range(len(sizes) - 2, -1, -1):
            if i == len(sizes) - 2:
                add('proj%d.WT' % i, sizes[i + 1], sizes[i]); add('proj%d.b' % i, sizes[i + 1])
            else:
                add('proj%d.W' % i, sizes[i] + 1, sizes[i + 1])
        add('dec.Wx', spec.dec_embed + 1, 4 * spec.dec_rnn)
        add('dec.Wh', 1, spec.dec_rnn, 4 * spec.dec_rnn)
        add('dec.emb', spec.vocab, spec.dec_embed)
        for l in range(len(spec.enc_rnn) - 1, -1, -1):
            Hh = spec.enc_rnn[l]
            if spec.aux_layer == l:
                asz = [2 * Hh] + list(spec.aux_hidden) + [spec.aux_dim]
                for i in range(len(asz) - 2, -1, -1):
                    if i == len(asz) - 2:
                        add('aux%d.WT' % i, asz[i + 1], asz[i]); add('aux%d.b' % i, asz[i + 1])
                    else:
                        add('aux%d.W' % i, asz[i] + 1, asz[i + 1])
            for j, hx in enumerate(spec.aux_extra):
                if hx['layer'] == l:
                    asz = [2 * Hh] + list(hx.get('hidden', [])) + [hx['dim']]
                    for i in range(len(asz) - 2, -1, -1):
                        if i == len(asz) - 2:
                            add('auxx%d_%d.WT' % (j, i), asz[i + 1], asz[i]); add('auxx%d_%d.b' % (j, i), asz[i + 1])
                        else:
                            add('auxx%d_%d.W' % (j, i), asz[i] + 1, asz[i + 1])
            D = spec.enc_embed if l == 0 else 2 * spec.enc_rnn[l - 1]
            add('enc%d.Wx' % l, D + 1, 2 * 4 * Hh)
            add('enc%d.Wh' % l, 2, Hh, 4 * Hh)
        self.shared_end = off
        for sid, Cc in spec.channels.items():
            lays = conv_stack(spec, Cc)
            for j in range(len(lays) - 1, -1, -1):          # top conv layer first: the order backward produces them in
                ci, co, n = lays[j]
                add(conv_seg(sid, j), n * ci + 1, co)
        self.n = off
        z = lambda: torch.zeros(self.n, dtype=torch.float32, device=device)
        self.p, self.g, self.m, self.v, self.ema = z(), z(), z(), z(), z()

    def view(self, name, buf=None):
        off, shape = self.segs[name]
        buf = self.p if buf is None else buf
        return buf[off:off + int(np.prod(shape))].view(*shape)

    def ptr(self, name, buf=None, elem_off=0):
        off, _ = self.segs[name]
        buf = self.p if buf is None else buf
        return buf.data_ptr() + 4 * (off + elem_off)

    def seg_range(self, name):
        off, shape = self.segs[name]
        return off, off + seg_pad(int(np.prod(shape)))

    # ---- TF-layout names (checkpoint grammar, trainers.py:444-554) -------------
    def tf_names(self):
        s = self.spec
        out = []
        for sid, Cc in s.channels.items():
            out.append('seq2seq/subnet_%s/encoder_embedding_%d_%d_0' % (sid, Cc, s.enc_embed))
        return out

    def import_tf(self, P, bufs=('p', 'ema')):
        """Load a dict of TF-layout arrays (oracle.init_params naming) into the masters."""
        s = self.spec
        N = s.decimation
        for bn in bufs:
            buf = getattr(self, bn)

            def put(name, arr):
                self.view(name, buf).copy_(torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.float32))
            for sid, Cc in s.channels.items():
                for j, (ci, co, n) in enumerate(conv_stack(s, Cc)):
                    nm = conv_tf_name(sid, j, ci, co)
                    put(conv_seg(sid, j), np.concatenate([P[nm + '/weights'].reshape(n * ci, co), P[nm + '/biases'][None]], 0))
            for l, Hh in enumerate(s.enc_rnn):
                D = s.enc_embed if l == 0 else 2 * s.enc_rnn[l - 1]
                wx, wh = [], []
                for d in ('fw', 'bw'):
                    K = P['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, d)]
                    b = P['seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, d)]
                    wx.append(np.concatenate([_tf2int(K[:D], Hh), _tf2int(b[None], Hh)], 0))
                    wh.append(_tf2int(K[D:], Hh))
                put('enc%d.Wx' % l, np.concatenate(wx, 1))
                put('enc%d.Wh' % l, np.stack(wh, 0))
            self._ff_io(P, 'aux', 'encoder_%s_projection' % s.aux_layer,
                        None if s.aux_layer is None else [2 * s.enc_rnn[s.aux_layer]] + list(s.aux_hidden) + [s.aux_dim], put)
            for j, hx in enumerate(s.aux_extra):
                self._ff_io(P, 'auxx%d_' % j, 'encoder_%s_projection' % hx['layer'],
                            [2 * s.enc_rnn[hx['layer']]] + list(hx.get('hidden', [])) + [hx['dim']], put)
            put('dec.emb', P['seq2seq/decoder_embedding_%d_%d_0/weights' % (s.vocab, s.dec_embed)])
            K = P['seq2seq/decoder_rnn/cell_0/kernel']
            b = P['seq2seq/decoder_rnn/cell_0/bias']
            put('dec.Wx', np.concatenate([_tf2int(K[:s.dec_embed], s.dec_rnn), _tf2int(b[None], s.dec_rnn)], 0))
            put('dec.Wh', _tf2int(K[s.dec_embed:], s.dec_rnn)[None])
            self._ff_io(P, 'proj', 'decoder_projection', [s.dec_rnn] + list(s.dec_proj_hidden) + [s.vocab], put)

    def _ff_io(self, P, prefix, tfprefix, sizes, put):
        if sizes is None:
            return
        for i in range(len(sizes) - 1):
            nm = 'seq2seq/%s_%d_%d_%d' % (tfprefix, sizes[i], sizes[i + 1], i)
            if i == len(sizes) - 2:
                put('%s%d.WT' % (prefix, i), P[nm + '/weights'])
                put('%s%d.b' % (prefix, i), P[nm + '/biases'])
            else:
                put('%s%d.W' % (prefix, i), np.concatenate([P[nm + '/weights'], P[nm + '/biases'][None]], 0))

    def export_tf(self, which='p'):
        """Inverse of import_tf: dict of TF-layout float64 numpy arrays."""
        s = self.spec
        N = s.decimation
        buf = getattr(self, which)
        host = buf.detach().cpu().numpy().astype(np.float64)

        def get(name):
            off, shape = self.segs[name]
            return host[off:off + int(np.prod(shape))].reshape(shape)
        out = {}
        for sid, Cc in s.channels.items():
            for j, (ci, co, n) in enumerate(conv_stack(s, Cc)):
                nm = conv_tf_name(sid, j, ci, co)
                w = get(conv_seg(sid, j))
                out[nm + '/weights'] = w[:-1].reshape(1, n, ci, co).copy()
                out[nm + '/biases'] = w[-1].copy()
        for l, Hh in enumerate(s.enc_rnn):
            D = s.enc_embed if l == 0 else 2 * s.enc_rnn[l - 1]
            wx, wh = get('enc%d.Wx' % l), get('enc%d.Wh' % l)
            for d, dn in enumerate(('fw', 'bw')):
                blk = wx[:, d * 4 * Hh:(d + 1) * 4 * Hh]
                out['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, dn)] = np.concatenate(
                    [_int2tf(blk[:D], Hh), _int2tf(wh[d], Hh)], 0)
                out['seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, dn)] = _int2tf(blk[D:D + 1], Hh)[0]
        if s.aux_layer is not None:
            self._ff_out(out, get, 'aux', 'encoder_%s_projection' % s.aux_layer,
                         [2 * s.enc_rnn[s.aux_layer]] + list(s.aux_hidden) + [s.aux_dim])
        for j, hx in enumerate(s.aux_extra):
            self._ff_out(out, get, 'auxx%d_' % j, 'encoder_%s_projection' % hx['layer'],
                         [2 * s.enc_rnn[hx['layer']]] + list(hx.get('hidden', [])) + [hx['dim']])
        out['seq2seq/decoder_embedding_%d_%d_0/weights' % (s.vocab, s.dec_embed)] = get('dec.emb').copy()
        wx, wh = get('dec.Wx'), get('dec.Wh')
        out['seq2seq/decoder_rnn/cell_0/kernel'] = np.concatenate(
            [_int2tf(wx[:s.dec_embed], s.dec_rnn), _int2tf(wh[0], s.dec_rnn)], 0)
        out['seq2seq/decoder_rnn/cell_0/bias'] = _int2tf(wx[s.dec_embed:], s.dec_rnn)[0]
        self._ff_out(out, get, 'proj', 'decoder_projection', [s.dec_rnn] + list(s.dec_proj_hidden) + [s.vocab])
        return out

    def _ff_out(self, out, get, prefix, tfprefix, sizes):
        for i in range(len(sizes) - 1):
            nm = 'seq2seq/%s_%d_%d_%d' % (tfprefix, sizes[i], sizes[i + 1], i)
            if i == len(sizes) - 2:
                out[nm + '/weights'] = get('%s%d.WT' % (prefix, i)).copy()
                out[nm + '/biases'] = get('%s%d.b' % (prefix, i)).copy()
            else:
                w = get('%s%d.W' % (prefix, i))
                out[nm + '/weights'] = w[:-1].copy()
                out[nm + '/biases'] = w[-1].copy()
