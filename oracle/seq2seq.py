"""NumPy fp64 restatement of the ECoG->text sequence-to-sequence network.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
arithmetic of this path lives in the un-vendored TF1.x dependency
`machine_learning.neural_networks.sequence_networks.SequenceNetwork`
(reference call sites: ecog2txt/trainers.py:126-135 ctor, :318/:355/:367 fit,
:379-380 restore_and_assess).  What the reference itself fixes, and this file
follows, is cited per function; every remaining choice is a [BUILD-DEFINES]
item recorded in DESIGN.md section "Normative spec".

All sequence tensors are TIME-MAJOR here ([S, B, ...]); the public batch is
batch-major [B, T, C] exactly as the reference's padded batches
(data_generators.py:247-315).  `q` is the bf16-rounding hook: identity for the
exact fp64 spec, oracle.bf16.round_bf16 to emulate the HIP path's operand
rounding points (DESIGN.md "Rounding points").
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .philox import keep_mask
from .bf16 import round_bf16

PAD_ID, EOS_ID, OOV_ID = 0, 1, 2      # special-token order, trainers.py:191-196

# dropout stream ids (shared with ecog2txt_amd/csrc/philox.h)
STREAM_CONV = 1          # the conv layer that feeds the encoder; earlier conv layers: STREAM_CONV_PRE + index
STREAM_CONV_PRE = 40
STREAM_ENC = 10          # + layer index
STREAM_DEC_EMB = 20
STREAM_DEC_OUT = 21
STREAM_AUX = 30


def _identity(x):
    return x


@dataclass
class NetSpec:
    """Sizes of one network (manifest keys: mocha-1_word_sequence.yaml:56-69)."""
    channels: Dict[object, int]            # subnet_id -> num ECoG channels (subjects.py:168-170)
    decimation: int = 12                   # subjects.py:144-153
    enc_embed: int = 100                   # layer_sizes['encoder_embedding'][0]
    enc_rnn: List[int] = field(default_factory=lambda: [400, 400, 400])
    dec_embed: int = 150
    dec_rnn: int = 800
    dec_proj_hidden: List[int] = field(default_factory=list)
    vocab: int = 1806
    aux_layer: Optional[int] = 1           # digit in 'encoder_1_targets' (trainers.py:788-799)
    aux_hidden: List[int] = field(default_factory=lambda: [225])
    aux_dim: int = 13
    aux_dist: str = 'Gaussian'             # subjects.py:369-380
    aux_scale: float = 1.0                 # '<data_key>_penalty_scale' trainers.py:98-102
    # further auxiliary heads, one per additional 'encoder_<k>_targets' data key (the reference loops over every data key:
    # trainers.py:94-102, 786-799): dicts with layer, hidden, dim, dist, scale; their targets come in
    # batch['encoder_targets_extra'][i]; Philox stream STREAM_AUX + 4 * (i + 1)
    aux_extra: List[dict] = field(default_factory=list)
    dec_scale: float = 1.0
    ff_dropout: float = 0.1
    rnn_dropout: float = 0.5
    forget_bias: float = 1.0
    conv_relu: bool = True
    # conv layers IN FRONT of the one that feeds the encoder (layer_sizes['encoder_embedding'] with more than one entry;
    # the reference only shows that the layers' strides multiply to decimation_factor and that width == stride,
    # trainers.py:406-407, 535-541): dicts with out, stride.  [BUILD-DEFINES] the split of decimation_factor over the layers is
    # given explicitly; the last layer (output enc_embed) gets decimation / prod(strides of conv_pre)
    conv_pre: List[dict] = field(default_factory=list)


# --------------------------------------------------------------------------
# parameter naming: checkpoint grammar of trainers.py:444-554
# --------------------------------------------------------------------------
def conv_layers(spec, sid):
    """[(name, in width, out width, stride)] of the subject's temporal-convolution stack, bottom up."""
    outs = [int(p['out']) for p in spec.conv_pre] + [spec.enc_embed]
    strides = [int(p['stride']) for p in spec.conv_pre]
    last = spec.decimation // int(np.prod(strides)) if strides else spec.decimation
    assert last * int(np.prod(strides or [1])) == spec.decimation, 'the strides must multiply to the decimation factor'
    strides.append(last)
    ins = [spec.channels[sid]] + outs[:-1]
    return [('seq2seq/subnet_%s/encoder_embedding_%d_%d_%d' % (sid, i, o, j), i, o, n) for j, (i, o, n) in enumerate(zip(ins, outs, strides))]


def conv_name(spec, sid):
    return conv_layers(spec, sid)[-1][0] if not spec.conv_pre else conv_layers(spec, sid)[0][0]


def enc_in_width(spec, l):
    return spec.enc_embed if l == 0 else 2 * spec.enc_rnn[l - 1]


def ff_names(prefix, sizes):
    """sizes = [in, h1, ..., out]; last layer is stored transposed (trainers.py:513-520)."""
    return ['seq2seq/%s_%d_%d_%d' % (prefix, sizes[i], sizes[i + 1], i) for i in range(len(sizes) - 1)]


def aux_heads(spec):
    """All auxiliary heads of a network: the primary one (aux_layer & co.) first, then aux_extra.  One head per tapped
    layer (the variable names carry the layer index only: 'encoder_<layer>_projection_...')."""
    heads = []
    if spec.aux_layer is not None:
        heads.append(dict(layer=spec.aux_layer, hidden=list(spec.aux_hidden), dim=spec.aux_dim, dist=spec.aux_dist,
                          scale=spec.aux_scale, key='encoder_targets', stream=STREAM_AUX, name='aux'))
    for i, h in enumerate(spec.aux_extra):
        heads.append(dict(layer=h['layer'], hidden=list(h.get('hidden', [])), dim=h['dim'], dist=h.get('dist', 'Gaussian'),
                          scale=h.get('scale', 1.0), key=i, stream=STREAM_AUX + 4 * (i + 1), name='aux_x%d' % i))
    assert len({h['layer'] for h in heads}) == len(heads), 'one auxiliary head per encoder layer'
    return heads


def init_params(spec, seed=0, dtype=np.float64):
    """Glorot-uniform weights, zero biases [BUILD-DEFINES]."""
    rng = np.random.default_rng(seed)
    P = {}

    def glorot(*shape, fan=None):
        fi, fo = fan if fan else (shape[0], shape[-1])
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    for sid in spec.channels:
        for nm, ci, co, n in conv_layers(spec, sid):
            P[nm + '/weights'] = glorot(1, n, ci, co, fan=(n * ci, co))
            P[nm + '/biases'] = np.zeros(co, dtype)
    for l, H in enumerate(spec.enc_rnn):
        D = enc_in_width(spec, l)
        for d in ('fw', 'bw'):
            P['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, d)] = glorot(D + H, 4 * H)
            P['seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, d)] = np.zeros(4 * H, dtype)
    for hd in aux_heads(spec):
        sizes = [2 * spec.enc_rnn[hd['layer']]] + list(hd['hidden']) + [hd['dim']]
        names = ff_names('encoder_%d_projection' % hd['layer'], sizes)
        for i, nm in enumerate(names):
            last = i == len(names) - 1
            w = glorot(sizes[i], sizes[i + 1])
            P[nm + '/weights'] = w.T.copy() if last else w
            P[nm + '/biases'] = np.zeros(sizes[i + 1], dtype)
    P['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)] = glorot(spec.vocab, spec.dec_embed)
    P['seq2seq/decoder_rnn/cell_0/kernel'] = glorot(spec.dec_embed + spec.dec_rnn, 4 * spec.dec_rnn)
    P['seq2seq/decoder_rnn/cell_0/bias'] = np.zeros(4 * spec.dec_rnn, dtype)
    sizes = [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab]
    names = ff_names('decoder_projection', sizes)
    for i, nm in enumerate(names):
        last = i == len(names) - 1
        w = glorot(sizes[i], sizes[i + 1])
        P[nm + '/weights'] = w.T.copy() if last else w
        P[nm + '/biases'] = np.zeros(sizes[i + 1], dtype)
    return P


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def sequence_lengths(x):
    """Valid length = number of non-zero rows of a zero-padded batch [B,T,C]
    (nn.sequences_tools usage at trainers.py:789-790, 806-807; padding value 0.0
    subjects.py:386-390)."""
    x = np.asarray(x)
    if x.ndim == 2:
        return (x != PAD_ID).sum(1).astype(np.int64)
    return (np.abs(x).max(axis=2) > 0).sum(1).astype(np.int64)


def reverse_time_major(x, lens):
    """tf.reverse_sequence(x, lens, seq_axis=1, batch_axis=0) (trainers.py:791-793,
    808-810) applied to batch-major x [B,T,...]; returns TIME-major [T,B,...] with
    zeros beyond each valid length."""
    B, T = x.shape[:2]
    out = np.zeros((T, B) + x.shape[2:], dtype=x.dtype)
    for b in range(B):
        n = int(lens[b])
        out[:n, b] = x[b, :n][::-1]
    return out


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def ceil_div(a, b):
    return -(-a // b)


def _drop(x, rate, seed, stream, train):
    """Inverted dropout with the shared Philox mask; returns (y, scale_mask)."""
    if not train or rate <= 0.0:
        return x, None
    m = keep_mask(x.shape, rate, seed, stream) / (1.0 - rate)
    return x * m, m


# --------------------------------------------------------------------------
# LSTM direction: TF1 LSTMCell convention [RECALL, SURVEY App. D1]: kernel
# [D+H, 4H], gate order i, j, f, o (4-gate packing: trainers.py:527-529),
# forget_bias added before the sigmoid; zero initial state for the encoder.
# Steps beyond an utterance's length carry the state and emit zeros.
# --------------------------------------------------------------------------
def lstm_dir_fwd(Gx, lens, Wh, reverse, q, forget_bias, h0=None, c0=None):
    S, B, H4 = Gx.shape
    H = H4 // 4
    rows = np.arange(B)
    Y = np.zeros((S, B, H))          # un-rounded h_t at its time index (0 on padding)
    Yq = np.zeros((S, B, H))         # q(h_t): what the recurrence and later GEMMs read
    Cs = np.zeros((S, B, H))
    Gs = np.zeros((S, B, 4, H))
    h = np.zeros((B, H)) if h0 is None else h0.copy()
    c = np.zeros((B, H)) if c0 is None else c0.copy()
    for s in range(S):
        act = s < lens
        t = np.where(reverse, lens - 1 - s, s)
        t = np.clip(t, 0, S - 1)
        z = Gx[t, rows] + h @ Wh                      # h is already q()'d state
        i = sigmoid(z[:, 0:H])
        j = np.tanh(z[:, H:2 * H])
        f = sigmoid(z[:, 2 * H:3 * H] + forget_bias)
        o = sigmoid(z[:, 3 * H:4 * H])
        cn = f * c + i * j
        hn = o * np.tanh(cn)
        a = act[:, None]
        ta, ra = t[act], rows[act]
        Y[ta, ra] = hn[act]
        Yq[ta, ra] = q(hn)[act]
        Cs[ta, ra] = cn[act]
        Gs[ta, ra] = q(np.stack([i, j, f, o], 1))[act]      # saved for BPTT in bf16 on the device (the forward pass uses them unrounded)
        c = np.where(a, cn, c)
        h = np.where(a, q(hn), h)
    return Y, Yq, dict(Cs=Cs, Gs=Gs, lens=lens, reverse=reverse, Wh=Wh, h0=h0, c0=c0,
                       Yq=Yq, hT=h, cT=c, H=H)


def lstm_dir_bwd(cache, dY, q, dh_final=None, dc_final=None):
    """BPTT for one direction.  dY [S,B,H]: gradient w.r.t. the un-dropped output
    h_t at each time index.  Returns dG (rounded, [S,B,4H]), dWh, dh0, dc0."""
    Cs, Gs, lens, reverse, Wh = cache['Cs'], cache['Gs'], cache['lens'], cache['reverse'], cache['Wh']
    Yq, h0, c0, H = cache['Yq'], cache['h0'], cache['c0'], cache['H']
    S, B = dY.shape[:2]
    rows = np.arange(B)
    dG = np.zeros((S, B, 4 * H))
    dc_carry = np.zeros((B, H))
    dWh = np.zeros_like(Wh)
    if dh_final is None:
        dh_final = np.zeros((B, H))
    if dc_final is None:
        dc_final = np.zeros((B, H))
    for s in range(S - 1, -1, -1):
        act = s < lens
        t = np.clip(np.where(reverse, lens - 1 - s, s), 0, S - 1)
        has_next = (s + 1) < lens
        t_next = np.clip(np.where(reverse, t - 1, t + 1), 0, S - 1)
        dGn = np.where(has_next[:, None], dG[t_next, rows], 0.0)
        rec = dGn @ Wh.T
        is_last = (s == lens - 1)
        dh = dY[t, rows] + rec + np.where(is_last[:, None], dh_final, 0.0)
        dc_in = np.where(has_next[:, None], dc_carry, dc_final)
        i, j, f, o = (Gs[t, rows, k] for k in range(4))
        c_t = Cs[t, rows]
        has_prev = s > 0
        t_prev = np.clip(np.where(reverse, t + 1, t - 1), 0, S - 1)
        if has_prev:
            c_prev = Cs[t_prev, rows]
            h_prev = Yq[t_prev, rows]
        else:
            c_prev = np.zeros((B, H)) if c0 is None else c0
            h_prev = np.zeros((B, H)) if h0 is None else h0
        tc = np.tanh(c_t)
        dct = dc_in + dh * o * (1.0 - tc * tc)
        do = dh * tc * o * (1.0 - o)
        di = dct * j * i * (1.0 - i)
        dj = dct * i * (1.0 - j * j)
        df = dct * c_prev * f * (1.0 - f)
        dg = q(np.concatenate([di, dj, df, do], 1))
        dg = np.where(act[:, None], dg, 0.0)
        ta, ra = t[act], rows[act]
        dG[ta, ra] = dg[act]
        dc_carry = np.where(act[:, None], dct * f, dc_carry)
        dWh += (h_prev * act[:, None]).T @ dg
    # gradient into the initial state (decoder): every utterance has s=0 active iff len>0
    act0 = lens > 0
    t0 = np.clip(np.where(reverse, lens - 1, 0), 0, S - 1)
    dh0 = np.where(act0[:, None], dG[t0, rows], 0.0) @ Wh.T + np.where(act0[:, None], 0.0, dh_final)
    dc0 = np.where(act0[:, None], dc_carry, dc_final)
    return dG, dWh, dh0, dc0


def _split_kernel(K, D):
    return K[:D], K[D:]


# --------------------------------------------------------------------------
# feed-forward head: hidden layers ReLU (+FF dropout), last layer linear with
# the weight stored transposed (trainers.py:513-520)
# --------------------------------------------------------------------------
def ff_fwd(P, names, x, spec, q, train, seed, stream):
    acts = [x]
    caches = []
    for i, nm in enumerate(names):
        last = i == len(names) - 1
        W = q(P[nm + '/weights'])
        W = W.T if last else W
        y = acts[-1] @ W + P[nm + '/biases']
        if not last:
            y = np.maximum(y, 0.0)
            y, m = _drop(y, spec.ff_dropout, seed, stream + i, train)
            y = q(y)
            caches.append(m)
        acts.append(y)
    return acts[-1], dict(acts=acts, masks=caches, names=names)


def ff_bwd(P, cache, dout, spec, q, grads):
    """dout is the (fp) gradient of the final linear output; returns d(input)."""
    names, acts, masks = cache['names'], cache['acts'], cache['masks']
    d = q(dout)
    for i in range(len(names) - 1, -1, -1):
        nm = names[i]
        last = i == len(names) - 1
        W = q(P[nm + '/weights'])
        x = acts[i]
        lead = x.reshape(-1, x.shape[-1])
        d2 = d.reshape(-1, d.shape[-1])
        if last:
            grads[nm + '/weights'] = d2.T @ lead          # stored transposed [out, in]
            dx = d @ W                                    # W is [out, in]
        else:
            grads[nm + '/weights'] = lead.T @ d2
            dx = d @ W.T
        grads[nm + '/biases'] = d2.sum(0)
        if i > 0:
            # x = acts[i] = q(dropout(relu(pre))): x > 0 iff relu active AND kept
            m = masks[i - 1]
            d = q(dx * (x > 0) * (m if m is not None else 1.0))
        else:
            d = dx
    return d


# --------------------------------------------------------------------------
# full forward
# --------------------------------------------------------------------------
def forward(P, spec, batch, train=False, seed=0, emulate_bf16=False, counts=None):
    """batch: dict(subnet_id, encoder_inputs [B,T,C], decoder_targets [B,L] int,
    optional encoder_targets [B,T,K] float (Gaussian) or [B,T] int (categorical)).
    counts = (tokens, auxiliary samples): normalise the losses by these instead of the batch's
    own counts (data parallel: the counts of the GLOBAL batch, so that the sum of the shards'
    gradients is the gradient of the global mean loss).
    Returns (losses dict, cache)."""
    q = round_bf16 if emulate_bf16 else _identity
    sid = batch['subnet_id']
    X = np.asarray(batch['encoder_inputs'], dtype=np.float64)
    B, T, C = X.shape
    N = spec.decimation
    lens = sequence_lengths(X)                        # a4
    Xr = reverse_time_major(X, lens)                  # a5, [T,B,C]
    S = ceil_div(T, N)
    lens_d = ceil_div(lens, N)
    # a6: strided temporal convolution(s), kernel width == stride (trainers.py:535-541, plotters.py:511-514), zero-padded
    # ragged tail; with conv_pre the layers are stacked, total stride N
    Xp = np.zeros((S * N, B, C))
    Xp[:T] = Xr
    cur, lens_cur, convs = Xp, lens, []
    layers = conv_layers(spec, sid)
    for j, (nm, ci, co, n) in enumerate(layers):
        Tj = cur.shape[0] // n
        Aj = q(cur.reshape(Tj, n, B, ci).transpose(0, 2, 1, 3).reshape(Tj, B, n * ci))
        Kj = q(P[nm + '/weights'].reshape(n * ci, co))
        Epre = Aj @ Kj + P[nm + '/biases']
        Eact = np.maximum(Epre, 0.0) if spec.conv_relu else Epre
        last = j == len(layers) - 1
        if last:
            Edrop, mE = _drop(Eact, spec.ff_dropout, seed, STREAM_CONV, train)
        else:
            # the mask is indexed in the order the device keeps a front layer's rows: (final step, utterance, step within
            # the group of g steps that ends up in one final step) -- what makes the next layer's im2row a plain view
            g = Tj // S
            Eg = Eact.reshape(S, g, B, co).transpose(0, 2, 1, 3)
            Edg, mg = _drop(Eg, spec.ff_dropout, seed, STREAM_CONV_PRE + j, train)
            Edrop = Edg.transpose(0, 2, 1, 3).reshape(Tj, B, co)
            mE = None if mg is None else mg.transpose(0, 2, 1, 3).reshape(Tj, B, co)
        lens_cur = ceil_div(lens_cur, n)
        vj = (np.arange(Tj)[:, None] < lens_cur[None, :])
        Ej = q(Edrop * vj[:, :, None])
        convs.append(dict(A=Aj, K=Kj, E=Ej, mE=mE, valid=vj, name=nm, n=n, ci=ci, co=co))
        cur = Ej
    E, A, Kc, mE, valid = cur, convs[0]['A'], convs[0]['K'], convs[-1]['mE'], convs[-1]['valid']
    assert np.array_equal(lens_cur, lens_d)
    cache = dict(spec=spec, batch=batch, q=q, train=train, seed=seed, lens=lens, lens_d=lens_d,
                 A=A, Kc=Kc, E=E, Eact=None, mE=mE, valid=valid, S=S, B=B, convs=convs)

    # a7: stacked bidirectional LSTM encoder
    inp = E
    enc = []
    for l, H in enumerate(spec.enc_rnn):
        D = inp.shape[-1]
        lay = dict(inp=inp)
        outs_raw, outs_q = [], []
        for d, name in enumerate(('fw', 'bw')):
            Kx, Kh = _split_kernel(P['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, name)], D)
            Wx, Wh = q(Kx), q(Kh)
            Gx = q(inp @ Wx + P['seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, name)])     # the device keeps the input projections in bf16
            Y, Yq, c = lstm_dir_fwd(Gx, lens_d, Wh, reverse=(d == 1), q=q, forget_bias=spec.forget_bias)
            c['Wx'] = Wx
            lay[name] = c
            outs_raw.append(Y)
            outs_q.append(Yq)
        Yraw = np.concatenate(outs_raw, -1)
        Ydrop, mY = _drop(Yraw, spec.rnn_dropout, seed, STREAM_ENC + l, train)
        Ydrop = q(Ydrop)
        lay.update(Ydrop=Ydrop, mY=mY, Yq=np.concatenate(outs_q, -1))
        enc.append(lay)
        inp = Ydrop
    cache['enc'] = enc
    last = enc[-1]
    h0 = np.concatenate([last['fw']['hT'], last['bw']['hT']], -1)      # App. D2
    c0 = np.concatenate([last['fw']['cT'], last['bw']['cT']], -1)
    cache.update(h0=h0, c0=c0)

    losses = {}
    # a8: auxiliary encoder-target heads (one per 'encoder_<k>_targets' data key)
    cache['aux_heads'] = {}
    for hi, hd in enumerate(aux_heads(spec)):
        if hd['key'] == 'encoder_targets':
            tg = batch.get('encoder_targets')
        else:
            ex = batch.get('encoder_targets_extra')
            tg = ex[hd['key']] if ex is not None else None
        if tg is None or hd['scale'] == 0.0:
            continue
        tg = np.asarray(tg)
        cat = hd['dist'] == 'categorical'
        tlens = sequence_lengths(tg)
        tr = reverse_time_major(tg if not cat else tg[..., None], tlens)     # [T,B,K]
        trp = np.zeros((S * N,) + tr.shape[1:], dtype=tr.dtype)
        trp[:T] = tr
        At = trp[0::N]                                   # trainers.py:794-795
        avalid = (np.arange(S)[:, None] * N < tlens[None, :])
        names = ff_names('encoder_%d_projection' % hd['layer'], [2 * spec.enc_rnn[hd['layer']]] + list(hd['hidden']) + [hd['dim']])
        Pout, ffc = ff_fwd(P, names, enc[hd['layer']]['Ydrop'], spec, q, train, seed, hd['stream'])
        if counts is None:
            nval = max(int(avalid.sum()), 1)
        else:
            nval = max(int(counts[1] if hd['name'] == 'aux' else counts[2 + hd['key']]), 1)
        if cat:
            ids = At[..., 0].astype(np.int64)
            mx = Pout.max(-1, keepdims=True)
            lse = mx[..., 0] + np.log(np.exp(Pout - mx).sum(-1))
            ce = lse - np.take_along_axis(Pout, ids[..., None], -1)[..., 0]
            losses[hd['name']] = float((ce * avalid).sum() / nval)
            prob = np.exp(Pout - lse[..., None])
            onehot = np.zeros_like(Pout)
            np.put_along_axis(onehot, ids[..., None], 1.0, -1)
            dP = (prob - onehot) * avalid[..., None] / nval
        else:
            diff = (Pout - At) * avalid[..., None]
            losses[hd['name']] = float((diff ** 2).sum() / (nval * hd['dim']))
            dP = 2.0 * diff / (nval * hd['dim'])
        rec = dict(ff=ffc, dP=dP, names=names, Pout=Pout, At=At, avalid=avalid, head=hd)
        cache['aux_heads'][hd['layer']] = rec
        if hd['name'] == 'aux':
            cache['aux'] = rec

    # a9: decoder, teacher-forced; <EOS> doubles as the start symbol [BUILD-DEFINES]
    Yt = np.asarray(batch['decoder_targets'], dtype=np.int64)            # [B,L]
    L = Yt.shape[1]
    dlens = sequence_lengths(Yt)
    U = np.concatenate([np.full((B, 1), EOS_ID, np.int64), Yt[:, :-1]], 1).T   # [L,B] inputs
    Emb = q(P['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)])
    e = Emb[U]
    e, mEm = _drop(e, spec.ff_dropout, seed, STREAM_DEC_EMB, train)
    e = q(e)
    Kx, Kh = _split_kernel(P['seq2seq/decoder_rnn/cell_0/kernel'], spec.dec_embed)
    Wx, Wh = q(Kx), q(Kh)
    Gx = q(e @ Wx + P['seq2seq/decoder_rnn/cell_0/bias'])
    Yd, Ydq, dc_ = lstm_dir_fwd(Gx, dlens, Wh, reverse=False, q=q, forget_bias=spec.forget_bias,
                                h0=h0, c0=c0)
    dc_['Wx'] = Wx
    Hdrop, mH = _drop(Yd, spec.rnn_dropout, seed, STREAM_DEC_OUT, train)
    Hdrop = q(Hdrop)
    pnames = ff_names('decoder_projection', [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab])
    logits, pffc = ff_fwd(P, pnames, Hdrop, spec, q, train, seed, STREAM_DEC_OUT + 1)
    tvalid = (np.arange(L)[:, None] < dlens[None, :])
    ntok = max(int(tvalid.sum()), 1) if counts is None else max(int(counts[0]), 1)
    mx = logits.max(-1, keepdims=True)
    lse = mx[..., 0] + np.log(np.exp(logits - mx).sum(-1))
    tgt = Yt.T
    ce = lse - np.take_along_axis(logits, tgt[..., None], -1)[..., 0]
    losses['decoder'] = float((ce * tvalid).sum() / ntok)
    prob = np.exp(logits - lse[..., None])
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, tgt[..., None], 1.0, -1)
    dlogits = (prob - onehot) * tvalid[..., None] / ntok
    pred = logits.argmax(-1)
    losses['accuracy'] = float(((pred == tgt) * tvalid).sum() / ntok)
    losses['total'] = spec.dec_scale * losses['decoder'] + sum(hd['scale'] * losses.get(hd['name'], 0.0) for hd in aux_heads(spec))
    cache.update(dec=dict(U=U, e=e, mEm=mEm, lstm=dc_, Hdrop=Hdrop, mH=mH, pff=pffc, pnames=pnames,
                          dlogits=dlogits, dlens=dlens, logits=logits, L=L, Ydq=Ydq))
    return losses, cache


# --------------------------------------------------------------------------
# full backward (manual reverse-mode), mirrors the HIP path's data flow
# --------------------------------------------------------------------------
def backward(P, cache):
    spec, q = cache['spec'], cache['q']
    sid = cache['batch']['subnet_id']
    S, B = cache['S'], cache['B']
    G = {}
    d = cache['dec']
    # decoder projection + CE
    dH = ff_bwd(P, d['pff'], d['dlogits'] * spec.dec_scale, spec, q, G)       # [L,B,Hd]
    if d['mH'] is not None:
        dH = dH * d['mH']
    lc = d['lstm']
    dG, dWh, dh0, dc0 = lstm_dir_bwd(lc, dH, q)
    L = d['L']
    e2 = d['e'].reshape(L * B, -1)
    dG2 = dG.reshape(L * B, -1)
    dWx = e2.T @ dG2
    G['seq2seq/decoder_rnn/cell_0/kernel'] = np.concatenate([dWx, dWh], 0)
    G['seq2seq/decoder_rnn/cell_0/bias'] = dG2.sum(0)
    de = dG @ lc['Wx'].T
    if d['mEm'] is not None:
        de = de * d['mEm']
    nmE = 'seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)
    dEmb = np.zeros_like(P[nmE], dtype=np.float64)
    np.add.at(dEmb, d['U'].reshape(-1), de.reshape(L * B, -1))
    G[nmE] = dEmb

    # encoder, top layer down
    enc = cache['enc']
    nl = len(enc)
    dYdrop = None
    for l in range(nl - 1, -1, -1):
        lay = enc[l]
        H = spec.enc_rnn[l]
        dYd = np.zeros((S, B, 2 * H)) if dYdrop is None else dYdrop
        if l in cache.get('aux_heads', {}):
            a = cache['aux_heads'][l]
            dYd = dYd + ff_bwd(P, a['ff'], a['dP'] * a['head']['scale'], spec, q, G)
        dYraw = dYd * lay['mY'] if lay['mY'] is not None else dYd
        inp = lay['inp']
        D = inp.shape[-1]
        dIn = np.zeros((S, B, D))
        for k, name in enumerate(('fw', 'bw')):
            c = lay[name]
            if l == nl - 1:
                dhf, dcf = dh0[:, k * H:(k + 1) * H], dc0[:, k * H:(k + 1) * H]
            else:
                dhf = dcf = None
            dGd, dWh_d, _, _ = lstm_dir_bwd(c, dYraw[..., k * H:(k + 1) * H], q, dhf, dcf)
            dG2 = dGd.reshape(S * B, -1)
            dWx_d = inp.reshape(S * B, -1).T @ dG2
            G['seq2seq/encoder_rnn_%d/%s/cell_0/kernel' % (l, name)] = np.concatenate([dWx_d, dWh_d], 0)
            G['seq2seq/encoder_rnn_%d/%s/cell_0/bias' % (l, name)] = dG2.sum(0)
            dIn = dIn + dGd @ c['Wx'].T
        dYdrop = dIn
    # conv front-end, top layer down
    dE = dYdrop
    for cv in reversed(cache['convs']):
        scale = cv['mE'] if cv['mE'] is not None else 1.0
        if spec.conv_relu:
            dEpre = dE * (cv['E'] > 0) * scale
        else:
            dEpre = dE * scale * cv['valid'][:, :, None]
        dEpre = q(dEpre)
        Tj = dEpre.shape[0]
        A2 = cv['A'].reshape(Tj * B, -1)
        d2 = dEpre.reshape(Tj * B, -1)
        G[cv['name'] + '/weights'] = (A2.T @ d2).reshape(P[cv['name'] + '/weights'].shape)
        G[cv['name'] + '/biases'] = d2.sum(0)
        if cv is not cache['convs'][0]:
            dA = dEpre @ cv['K'].T                                           # [Tj, B, n*ci]
            dE = dA.reshape(Tj, B, cv['n'], cv['ci']).transpose(0, 2, 1, 3).reshape(Tj * cv['n'], B, cv['ci'])
    cache['dEpre'] = dEpre                 # what input_gradient() back-projects
    return G


def input_gradient(P, cache):
    """a12 saliency support: d loss / d encoder_inputs, batch-major [B,T,C], after backward()
    (restore_and_get_saliencies, trainers.py:703-732: the penalty-weighted loss is back-propagated
    into the `encoder_inputs` placeholder; plotters.py:534-560 then takes norms over it).
    Chain: dEpre (what backward() left at the conv pre-activation) -> dA = dEpre . K^T through the
    strided convolution (kernel width == stride, trainers.py:535-541) -> un-im2row -> undo
    tf.reverse_sequence over each utterance's valid length (trainers.py:808-810).
    [BUILD-DEFINES] padding samples (t >= len) are not inputs: their gradient is defined as 0."""
    spec = cache['spec']
    B = cache['B']
    X = np.asarray(cache['batch']['encoder_inputs'])
    T, C = X.shape[1], X.shape[2]
    N = cache['convs'][0]['n']                                           # the bottom layer's stride (= decimation without conv_pre)
    S = cache['dEpre'].shape[0]
    dA = cache['dEpre'] @ cache['Kc'].T                                  # [S,B,N*C], K as the forward pass rounded it
    dXr = dA.reshape(S, B, N, C).transpose(0, 2, 1, 3).reshape(S * N, B, C)     # reversed-time, time-major
    dX = np.zeros((B, T, C))
    for b in range(B):
        n = int(cache['lens'][b])
        dX[b, :n] = dXr[:n, b][::-1]
    return dX


# --------------------------------------------------------------------------
# a10: Adam (TF1 AdamOptimizer formulation) + EMA shadows
# --------------------------------------------------------------------------
def adam_ema_step(P, G, state, lr=5e-4, b1=0.9, b2=0.999, eps=1e-8, ema_decay=0.99, trainable=None):
    t = state.setdefault('t', 0) + 1
    state['t'] = t
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m, v, ema = state.setdefault('m', {}), state.setdefault('v', {}), state.setdefault('ema', {})
    for k in P:
        if k not in ema:
            ema[k] = P[k].copy()
        if k not in G or (trainable is not None and not trainable(k)):
            continue
        g = G[k]
        m[k] = b1 * m.get(k, 0.0) + (1 - b1) * g
        v[k] = b2 * v.get(k, 0.0) + (1 - b2) * g * g
        P[k] = P[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)
        ema[k] = ema_decay * ema[k] + (1.0 - ema_decay) * P[k]
    return P, state


# --------------------------------------------------------------------------
# greedy decoding (beam_width: 1, mocha-1_word_sequence.yaml:31)
# --------------------------------------------------------------------------
def greedy_decode(P, spec, batch, max_len=20, emulate_bf16=False):
    q = round_bf16 if emulate_bf16 else _identity
    dummy = dict(batch)
    B = np.asarray(batch['encoder_inputs']).shape[0]
    dummy['decoder_targets'] = np.full((B, 1), EOS_ID, np.int64)
    dummy.pop('encoder_targets', None)
    _, cache = forward(P, spec, dummy, train=False, emulate_bf16=emulate_bf16)
    h, c = cache['h0'], cache['c0']
    Emb = q(P['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)])
    Kx, Kh = _split_kernel(P['seq2seq/decoder_rnn/cell_0/kernel'], spec.dec_embed)
    Wx, Wh = q(Kx), q(Kh)
    bias = P['seq2seq/decoder_rnn/cell_0/bias']
    pnames = ff_names('decoder_projection', [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab])
    H = spec.dec_rnn
    u = np.full(B, EOS_ID, np.int64)
    done = np.zeros(B, bool)
    out = np.full((B, max_len), PAD_ID, np.int64)
    all_logits = []
    for l in range(max_len):
        z = q(q(Emb[u]) @ Wx + bias) + h @ Wh
        i = sigmoid(z[:, :H]); j = np.tanh(z[:, H:2 * H])
        f = sigmoid(z[:, 2 * H:3 * H] + spec.forget_bias); o = sigmoid(z[:, 3 * H:])
        c = f * c + i * j
        hn = o * np.tanh(c)
        h = q(hn)
        logits, _ = ff_fwd(P, pnames, h, spec, q, False, 0, 0)
        all_logits.append(logits)
        tok = logits.argmax(-1)
        out[:, l] = np.where(done, PAD_ID, tok)
        done = done | (tok == EOS_ID)
        u = tok
        if done.all():
            break
    return out, np.stack(all_logits, 0)


def beam_decode(P, spec, batch, beam_width, max_len=20, temperature=1.0, emulate_bf16=False):
    """Beam search over the decoder (`beam_width`, mocha-1_word_sequence.yaml:31; `temperature`, :82 -- both are manifest keys
    the reference hands to the absent SequenceNetwork, so the exact rule is [BUILD-DEFINES]): the standard one of
    tf.contrib.seq2seq.BeamSearchDecoder with length_penalty_weight 0.
      * score of a hypothesis = sum of log softmax(logits / temperature) of its tokens;
      * at every step the W best of the W x V continuations survive (ties: lower beam, then lower token id);
        a hypothesis that has emitted <EOS> is finished: its only continuation is itself, score unchanged;
      * start: beam 0 holds <EOS> as start symbol with score 0, the other beams -inf (W copies of one state);
      * result: the best-scoring hypothesis per utterance after max_len steps, padded with <pad> behind its <EOS>.
    beam_width 1 is greedy decoding.  Returns (tokens [B, max_len], scores [B, W] of the final beams, best first)."""
    q = round_bf16 if emulate_bf16 else _identity
    W = int(beam_width)
    dummy = dict(batch)
    B = np.asarray(batch['encoder_inputs']).shape[0]
    dummy['decoder_targets'] = np.full((B, 1), EOS_ID, np.int64)
    dummy.pop('encoder_targets', None)
    _, cache = forward(P, spec, dummy, train=False, emulate_bf16=emulate_bf16)
    Emb = q(P['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)])
    Kx, Kh = _split_kernel(P['seq2seq/decoder_rnn/cell_0/kernel'], spec.dec_embed)
    Wx, Wh = q(Kx), q(Kh)
    bias = P['seq2seq/decoder_rnn/cell_0/bias']
    pnames = ff_names('decoder_projection', [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab])
    H, V = spec.dec_rnn, spec.vocab
    h = np.repeat(cache['h0'], W, axis=0)                     # row b*W + w
    c = np.repeat(cache['c0'], W, axis=0)
    u = np.full(B * W, EOS_ID, np.int64)
    score = np.full((B, W), -np.inf)
    score[:, 0] = 0.0
    done = np.zeros((B, W), bool)
    toks = np.full((B, W, max_len), PAD_ID, np.int64)
    for l in range(max_len):
        z = q(q(Emb[u]) @ Wx + bias) + h @ Wh
        i = sigmoid(z[:, :H]); j = np.tanh(z[:, H:2 * H])
        f = sigmoid(z[:, 2 * H:3 * H] + spec.forget_bias); o = sigmoid(z[:, 3 * H:])
        cn = f * c + i * j
        hn = q(o * np.tanh(cn))
        logits, _ = ff_fwd(P, pnames, hn, spec, q, False, 0, 0)
        x = logits / temperature
        x = x - x.max(-1, keepdims=True)
        logp = (x - np.log(np.exp(x).sum(-1, keepdims=True))).reshape(B, W, V)
        cand = score[:, :, None] + logp                          # [B, W, V]
        # a finished hypothesis continues only as itself (token <pad>, score unchanged)
        cand = np.where(done[:, :, None], -np.inf, cand)
        keep = np.where(done, score, -np.inf)                    # [B, W]: the "stay finished" candidate of each beam
        flat = np.concatenate([cand.reshape(B, W * V), keep], axis=1)      # candidate index: w*V + v, or W*V + w
        # order: best score first; ties by candidate index of (beam, token) with the stay-candidates ordered by their beam
        key_beam = np.concatenate([np.repeat(np.arange(W), V), np.arange(W)])
        key_tok = np.concatenate([np.tile(np.arange(V), W), np.full(W, -1)])
        new_score = np.empty((B, W)); parent = np.empty((B, W), np.int64); tok = np.empty((B, W), np.int64)
        for b in range(B):
            order = np.lexsort((key_tok, key_beam, -flat[b]))[:W]
            new_score[b] = flat[b, order]; parent[b] = key_beam[order]; tok[b] = key_tok[order]
        rows = (np.arange(B)[:, None] * W + parent).reshape(-1)
        h, c = hn[rows], cn[rows]
        toks = np.take_along_axis(toks, parent[:, :, None], axis=1)
        was_done = np.take_along_axis(done, parent, axis=1)
        stay = tok < 0
        toks[:, :, l] = np.where(stay | was_done, PAD_ID, tok)
        done = was_done | stay | (tok == EOS_ID)
        score = new_score
        u = np.where(stay, EOS_ID, tok).reshape(-1)             # (input of a finished beam: irrelevant)
        if done.all():
            break
    best = np.argmax(score, axis=1)                              # (beams are kept best first: index 0)
    return toks[np.arange(B), best], score
