"""Second, INDEPENDENT implementation of the ECoG->text network on torch-CPU (torch.nn.LSTM over packed sequences,
F.conv1d, F.cross_entropy, autograd) -- a different formulation with a different gate order and bias convention.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two uses:
  * tests/test_oracle_vs_torch.py pins the NumPy oracle against it in fp64 (losses, logits, every gradient, the input
    gradient) -- the reference ships no tests or golden vectors for this path (parity unpinned);
  * bench.py's `cpu_baseline` times it (fp32, torch's oneDNN/MKL LSTM kernels on all host cores, Adam + EMA) as the
    CPU number SURVEY.md 8 d5 (ii) asks for: the reference's own TF1 CPU path cannot run in this environment.
Reference call sites of what it restates: ecog2txt/trainers.py:126-135 (ctor), :318 (fit), :786-823 (forward fragment).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import seq2seq as O


def tf_to_torch_lstm(kernel, bias, D, H, forget_bias):
    """TF [D+H,4H] i,j,f,o  ->  torch weight_ih [4H,D], weight_hh [4H,H] in i,f,g,o."""
    Kx, Kh = kernel[:D], kernel[D:]

    def perm(w):     # columns i,j,f,o -> rows i,f,j,o
        i, j, f, o = (w[..., k * H:(k + 1) * H] for k in range(4))
        return torch.cat([i, f, j, o], -1)
    b = perm(bias[None])[0].clone()
    fb = torch.zeros(4 * H, dtype=bias.dtype)
    fb[H:2 * H] = forget_bias
    return perm(Kx).T, perm(Kh).T, b + fb


def run_lstm(x_tm, lens, kernel, bias, H, forget_bias, reverse_pair=None, h0=None, c0=None):
    """One uni-directional LSTM over time-major x using torch's fused _VF.lstm via nn.LSTM
    functional form (packed sequences give masking + final state at each row's own end)."""
    D = x_tm.shape[-1]
    wih, whh, b = tf_to_torch_lstm(kernel, bias, D, H, forget_bias)
    lstm = torch.nn.LSTM(D, H, batch_first=False)
    # functional call so autograd flows into OUR leaf tensors
    params = {'weight_ih_l0': wih, 'weight_hh_l0': whh, 'bias_ih_l0': b, 'bias_hh_l0': torch.zeros_like(b)}
    packed = pack_padded_sequence(x_tm, lens.cpu(), enforce_sorted=False)
    hx = None if h0 is None else (h0[None], c0[None])
    out, (hn, cn) = torch.func.functional_call(lstm, params, (packed, hx))
    out, _ = pad_packed_sequence(out, total_length=x_tm.shape[0])
    return out, hn[0], cn[0]


def reverse_padded(x_tm, lens):
    S, B = x_tm.shape[:2]
    idx = torch.arange(S)[:, None].expand(S, B)
    src = torch.where(idx < lens[None], lens[None] - 1 - idx, idx)
    return torch.gather(x_tm, 0, src[..., None].expand_as(x_tm))


def torch_model(Pt, spec, batch, masks, x_leaf=None):
    sid = batch['subnet_id']
    X = torch.tensor(batch['encoder_inputs']) if x_leaf is None else x_leaf
    B, T, C = X.shape
    N = spec.decimation
    lens = (X.abs().amax(2) > 0).sum(1)
    S = -(-T // N)
    lens_d = -(-lens // N)
    Xr = reverse_padded(X.transpose(0, 1), lens)                     # [T,B,C]
    Xr = F.pad(Xr, (0, 0, 0, 0, 0, S * N - T))
    # independent formulation of the temporal-convolution stack (one layer unless spec.conv_pre): F.conv1d over [B,C,T] with
    # stride = width per layer; the decimated lengths follow by ceil-division layer by layer
    cur, lens_j = Xr, lens
    layers = O.conv_layers(spec, sid)
    for j, (nm, ci, co, n) in enumerate(layers):
        W = Pt[nm + '/weights'][0]                                    # [n,ci,co]
        y = F.conv1d(cur.permute(1, 2, 0), W.permute(2, 1, 0), Pt[nm + '/biases'], stride=n)  # [B,co,Tj]
        E = y.permute(2, 0, 1)
        if spec.conv_relu:
            E = F.relu(E)
        mk = masks.get('conv' if j == len(layers) - 1 else 'conv_pre%d' % j)
        if mk is not None:
            E = E * mk
        lens_j = -(-lens_j // n)
        vj = (torch.arange(E.shape[0])[:, None] < lens_j[None]).to(E.dtype)
        E = E * vj[..., None]
        cur = E
    valid = (torch.arange(S)[:, None] < lens_d[None]).to(E.dtype)
    inp = E
    lens_c = torch.clamp(lens_d, min=1)
    taps = []
    for l, H in enumerate(spec.enc_rnn):
        k = 'seq2seq/encoder_rnn_%d/%s/cell_0/' % (l, '%s')
        of, hf, cf = run_lstm(inp, lens_c, Pt[(k % 'fw') + 'kernel'], Pt[(k % 'fw') + 'bias'], H, spec.forget_bias)
        ob, hb, cb = run_lstm(reverse_padded(inp, lens_d), lens_c, Pt[(k % 'bw') + 'kernel'],
                              Pt[(k % 'bw') + 'bias'], H, spec.forget_bias)
        ob = reverse_padded(ob, lens_d)
        Y = torch.cat([of, ob], -1) * valid[..., None]
        if masks.get('enc%d' % l) is not None:
            Y = Y * masks['enc%d' % l]
        taps.append(Y)
        inp = Y
    h0 = torch.cat([hf, hb], -1)
    c0 = torch.cat([cf, cb], -1)
    total = 0.0
    out = {}
    for hd in O.aux_heads(spec):
        if hd['key'] == 'encoder_targets':
            tgn = batch.get('encoder_targets')
        else:
            tgn = (batch.get('encoder_targets_extra') or [None] * (hd['key'] + 1))[hd['key']]
        if tgn is None or hd['scale'] == 0.0:
            continue
        tg = torch.tensor(np.asarray(tgn))
        cat = hd['dist'] == 'categorical'
        tl = (tg != 0).sum(1) if cat else (tg.abs().amax(2) > 0).sum(1)
        tgm = tg.transpose(0, 1)
        if cat:
            tgm = tgm[..., None]
        tr = reverse_padded(tgm, tl)
        tr = F.pad(tr, (0, 0, 0, 0, 0, S * N - T))[0::N]
        av = (torch.arange(S)[:, None] * N < tl[None]).to(E.dtype)
        sizes = [2 * spec.enc_rnn[hd['layer']]] + list(hd['hidden']) + [hd['dim']]
        names = O.ff_names('encoder_%d_projection' % hd['layer'], sizes)
        mkey = 'aux%d' if hd['name'] == 'aux' else ('auxx%d_' % hd['key']) + '%d'
        z = taps[hd['layer']]
        for i, n_ in enumerate(names):
            last = i == len(names) - 1
            if last:
                z = F.linear(z, Pt[n_ + '/weights'], Pt[n_ + '/biases'])
            else:
                z = F.relu(z @ Pt[n_ + '/weights'] + Pt[n_ + '/biases'])
                if masks.get(mkey % i) is not None:
                    z = z * masks[mkey % i]
        nval = av.sum().clamp(min=1)
        if cat:
            ce = F.cross_entropy(z.reshape(S * B, -1), tr[..., 0].reshape(-1).long(), reduction='none')
            aux = (ce * av.reshape(-1)).sum() / nval
        else:
            aux = (((z - tr) * av[..., None]) ** 2).sum() / (nval * hd['dim'])
        out[hd['name']] = aux
        total = total + hd['scale'] * aux
    Yt = torch.tensor(np.asarray(batch['decoder_targets'])).long()
    L = Yt.shape[1]
    dl = (Yt != 0).sum(1)
    U = torch.cat([torch.full((B, 1), O.EOS_ID), Yt[:, :-1]], 1).T
    e = Pt['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)][U]
    if masks.get('demb') is not None:
        e = e * masks['demb']
    od, _, _ = run_lstm(e, dl, Pt['seq2seq/decoder_rnn/cell_0/kernel'], Pt['seq2seq/decoder_rnn/cell_0/bias'],
                        spec.dec_rnn, spec.forget_bias, h0=h0, c0=c0)
    if masks.get('dout') is not None:
        od = od * masks['dout']
    pn = O.ff_names('decoder_projection', [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab])
    z = od
    for i, n_ in enumerate(pn):
        if i == len(pn) - 1:
            z = F.linear(z, Pt[n_ + '/weights'], Pt[n_ + '/biases'])
        else:
            z = F.relu(z @ Pt[n_ + '/weights'] + Pt[n_ + '/biases'])
            if masks.get('proj%d' % i) is not None:
                z = z * masks['proj%d' % i]
    tv = (torch.arange(L)[:, None] < dl[None]).to(E.dtype)
    ce = F.cross_entropy(z.reshape(L * B, -1), Yt.T.reshape(-1), reduction='none')
    dec = (ce * tv.reshape(-1)).sum() / tv.sum()
    out['decoder'] = dec
    out['total'] = total + spec.dec_scale * dec
    out['logits'] = z
    return out


def train_step_fn(spec, batch, lr=5e-4, ema_decay=0.99, dtype=torch.float32, seed=0):
    """A closure running ONE full optimisation step (forward, losses, backward, Adam, EMA) of the architecture `spec`
    on `batch` with torch-CPU kernels; dropout masks are drawn by torch (Bernoulli), not Philox -- this model is timed,
    not compared, when it trains.  Returns (step, params)."""
    P = O.init_params(spec, seed=seed)
    Pt = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in P.items()}
    ema = {k: v.detach().clone() for k, v in Pt.items()}
    opt = torch.optim.Adam(list(Pt.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8)
    b = dict(batch)
    b['encoder_inputs'] = np.asarray(batch['encoder_inputs'], dtype=np.float32 if dtype == torch.float32 else np.float64)
    if 'encoder_targets' in b and np.asarray(b['encoder_targets']).dtype.kind == 'f':
        b['encoder_targets'] = np.asarray(b['encoder_targets'], dtype=b['encoder_inputs'].dtype)
    X = b['encoder_inputs']
    B, T, _ = X.shape
    S = -(-T // spec.decimation)
    L = np.asarray(b['decoder_targets']).shape[1]
    gen = torch.Generator().manual_seed(seed)

    def mask(shape, rate):
        if rate <= 0:
            return None
        return (torch.rand(shape, generator=gen) >= rate).to(dtype) / (1.0 - rate)

    def step():
        masks = {'conv': mask((S, B, spec.enc_embed), spec.ff_dropout), 'demb': mask((L, B, spec.dec_embed), spec.ff_dropout),
                 'dout': mask((L, B, spec.dec_rnn), spec.rnn_dropout)}
        for l, H in enumerate(spec.enc_rnn):
            masks['enc%d' % l] = mask((S, B, 2 * H), spec.rnn_dropout)
        for i, h in enumerate(spec.aux_hidden):
            masks['aux%d' % i] = mask((S, B, h), spec.ff_dropout)
        for i, h in enumerate(spec.dec_proj_hidden):
            masks['proj%d' % i] = mask((L, B, h), spec.ff_dropout)
        opt.zero_grad(set_to_none=True)
        out = torch_model(Pt, spec, b, masks)
        out['total'].backward()
        opt.step()
        with torch.no_grad():
            for k, v in Pt.items():
                ema[k].mul_(ema_decay).add_(v, alpha=1.0 - ema_decay)
        return float(out["total"].detach())
    return step, Pt
