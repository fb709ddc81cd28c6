"""Pin the NumPy oracle against an INDEPENDENT torch-CPU autograd model.

The reference ships no tests or golden vectors for this path (SURVEY.md 8c:
parity unpinned), so the oracle's exact (fp64) mode is cross-checked here
against torch.nn.LSTM / packed sequences / F.cross_entropy -- a different
implementation with a different gate order and bias convention.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from oracle import seq2seq as O
from helpers import tiny_spec, make_batch



@pytest.fixture(autouse=True)
def _float64_default():
    """fp64 torch model; restore the default so other test modules are unaffected."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def tf_to_torch_lstm(kernel, bias, D, H, forget_bias):
    """TF [D+H,4H] i,j,f,o  ->  torch weight_ih [4H,D], weight_hh [4H,H] in i,f,g,o."""
    Kx, Kh = kernel[:D], kernel[D:]

    def perm(w):     # columns i,j,f,o -> rows i,f,j,o
        i, j, f, o = (w[..., k * H:(k + 1) * H] for k in range(4))
        return torch.cat([i, f, j, o], -1)
    b = perm(bias[None])[0].clone()
    fb = torch.zeros(4 * H)
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


def torch_model(Pt, spec, batch, masks):
    sid = batch['subnet_id']
    X = torch.tensor(batch['encoder_inputs'])
    B, T, C = X.shape
    N = spec.decimation
    lens = (X.abs().amax(2) > 0).sum(1)
    S = -(-T // N)
    lens_d = -(-lens // N)
    Xr = reverse_padded(X.transpose(0, 1), lens)                     # [T,B,C]
    Xr = F.pad(Xr, (0, 0, 0, 0, 0, S * N - T))
    nm = O.conv_name(spec, sid)
    W = Pt[nm + '/weights'][0]                                        # [N,C,F]
    # independent formulation: conv1d over [B,C,T] with stride N
    y = F.conv1d(Xr.permute(1, 2, 0), W.permute(2, 1, 0), Pt[nm + '/biases'], stride=N)  # [B,F,S]
    E = y.permute(2, 0, 1)
    if spec.conv_relu:
        E = F.relu(E)
    if masks.get('conv') is not None:
        E = E * masks['conv']
    valid = (torch.arange(S)[:, None] < lens_d[None]).to(E.dtype)
    E = E * valid[..., None]
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
    if 'encoder_targets' in batch and spec.aux_layer is not None:
        tg = torch.tensor(batch['encoder_targets'])
        cat = spec.aux_dist == 'categorical'
        tl = (tg != 0).sum(1) if cat else (tg.abs().amax(2) > 0).sum(1)
        tgm = tg.transpose(0, 1)
        if cat:
            tgm = tgm[..., None]
        tr = reverse_padded(tgm, tl)
        tr = F.pad(tr, (0, 0, 0, 0, 0, S * N - T))[0::N]
        av = (torch.arange(S)[:, None] * N < tl[None]).to(E.dtype)
        sizes = [2 * spec.enc_rnn[spec.aux_layer]] + list(spec.aux_hidden) + [spec.aux_dim]
        names = O.ff_names('encoder_%d_projection' % spec.aux_layer, sizes)
        z = taps[spec.aux_layer]
        for i, n_ in enumerate(names):
            last = i == len(names) - 1
            if last:
                z = F.linear(z, Pt[n_ + '/weights'], Pt[n_ + '/biases'])
            else:
                z = F.relu(z @ Pt[n_ + '/weights'] + Pt[n_ + '/biases'])
                if masks.get('aux%d' % i) is not None:
                    z = z * masks['aux%d' % i]
        nval = av.sum().clamp(min=1)
        if cat:
            ce = F.cross_entropy(z.reshape(S * B, -1), tr[..., 0].reshape(-1).long(), reduction='none')
            aux = (ce * av.reshape(-1)).sum() / nval
        else:
            aux = (((z - tr) * av[..., None]) ** 2).sum() / (nval * spec.aux_dim)
        out['aux'] = aux
        total = total + spec.aux_scale * aux
    Yt = torch.tensor(batch['decoder_targets'])
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


def collect_masks(cache, spec):
    """Hand the oracle's Philox masks to the torch model (masks are inputs, not under test)."""
    t = lambda m: None if m is None else torch.tensor(m)
    masks = {'conv': t(cache['mE']), 'demb': t(cache['dec']['mEm']), 'dout': t(cache['dec']['mH'])}
    for l, lay in enumerate(cache['enc']):
        masks['enc%d' % l] = t(lay['mY'])
    if 'aux' in cache:
        for i, m in enumerate(cache['aux']['ff']['masks']):
            masks['aux%d' % i] = t(m)
    for i, m in enumerate(cache['dec']['pff']['masks']):
        masks['proj%d' % i] = t(m)
    return masks


CASES = [
    dict(),
    dict(ff_dropout=0.2, rnn_dropout=0.5),
    dict(aux_dist='categorical', aux_dim=7),
    dict(aux_layer=None),
    dict(dec_proj_hidden=[9], ff_dropout=0.3),
    dict(conv_relu=False, enc_rnn=[4], aux_layer=0, dec_rnn=8),
    dict(enc_rnn=[4, 6, 5], dec_rnn=10, aux_layer=2),
]


@pytest.mark.parametrize('kw', CASES)
@pytest.mark.parametrize('ragged', [False, True])
def test_oracle_matches_torch_autograd(kw, ragged):
    spec = tiny_spec(**kw)
    P = O.init_params(spec, seed=3)
    rng = np.random.default_rng(5)
    for k in P:                      # non-zero biases so they are exercised
        if P[k].ndim == 1:
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    batch = make_batch(spec, B=5, T=11, L=6, seed=2, ragged=ragged, categorical=spec.aux_dist == 'categorical')
    train = spec.ff_dropout > 0 or spec.rnn_dropout > 0
    losses, cache = O.forward(P, spec, batch, train=train, seed=77)
    G = O.backward(P, cache)
    Pt = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
    out = torch_model(Pt, spec, batch, collect_masks(cache, spec))
    out['total'].backward()
    assert abs(losses['decoder'] - out['decoder'].item()) < 1e-10
    if 'aux' in out:
        assert abs(losses['aux'] - out['aux'].item()) < 1e-10
    assert abs(losses['total'] - out['total'].item()) < 1e-10
    np.testing.assert_allclose(cache['dec']['logits'], out['logits'].detach().numpy(), atol=1e-10)
    for k in P:
        g = Pt[k].grad
        if g is None:
            assert k not in G or np.abs(G[k]).max() == 0, k
            continue
        assert k in G, k
        np.testing.assert_allclose(G[k], g.numpy(), atol=1e-9, rtol=1e-8, err_msg=k)
