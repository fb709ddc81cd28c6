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


from oracle.torch_model import tf_to_torch_lstm, run_lstm, reverse_padded, torch_model      # noqa: E402,F401


def collect_masks(cache, spec):
    """Hand the oracle's Philox masks to the torch model (masks are inputs, not under test)."""
    t = lambda m: None if m is None else torch.tensor(m)
    masks = {'conv': t(cache['mE']), 'demb': t(cache['dec']['mEm']), 'dout': t(cache['dec']['mH'])}
    for l, lay in enumerate(cache['enc']):
        masks['enc%d' % l] = t(lay['mY'])
    for rec in cache.get('aux_heads', {}).values():
        key = 'aux%d' if rec['head']['name'] == 'aux' else ('auxx%d_' % rec['head']['key']) + '%d'
        for i, m in enumerate(rec['ff']['masks']):
            masks[key % i] = t(m)
    for j, cv in enumerate(cache['convs'][:-1]):
        masks['conv_pre%d' % j] = t(cv['mE'])
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
    # a stack of strided conv layers (strides 2 x 3 x 1 = decimation 6)
    dict(decimation=6, conv_pre=[dict(out=7, stride=2), dict(out=4, stride=3)], enc_embed=5, ff_dropout=0.2, rnn_dropout=0.1),
    dict(decimation=4, conv_pre=[dict(out=3, stride=4)], enc_embed=5),
    # several auxiliary heads (one per tapped layer): categorical on layer 0 next to the Gaussian one on layer 1
    dict(aux_extra=[dict(layer=0, hidden=[5], dim=4, dist='categorical', scale=0.7)], ff_dropout=0.2, rnn_dropout=0.3),
    dict(enc_rnn=[4, 6, 8], dec_rnn=16, aux_layer=1, aux_extra=[dict(layer=2, hidden=[], dim=3, scale=0.25), dict(layer=0, hidden=[6, 5], dim=2)]),
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
    for k in losses:
        if k.startswith('aux'):
            assert abs(losses[k] - out[k].item()) < 1e-10, k
    assert abs(losses['total'] - out['total'].item()) < 1e-10
    np.testing.assert_allclose(cache['dec']['logits'], out['logits'].detach().numpy(), atol=1e-10)
    for k in P:
        g = Pt[k].grad
        if g is None:
            assert k not in G or np.abs(G[k]).max() == 0, k
            continue
        assert k in G, k
        np.testing.assert_allclose(G[k], g.numpy(), atol=1e-9, rtol=1e-8, err_msg=k)



@pytest.mark.parametrize('kw', [dict(), dict(ff_dropout=0.2, rnn_dropout=0.5), dict(conv_relu=False, enc_rnn=[4], aux_layer=0, dec_rnn=8),
                                dict(decimation=6, conv_pre=[dict(out=7, stride=2), dict(out=4, stride=3)], enc_embed=5, ff_dropout=0.2)])
def test_oracle_input_gradient_matches_torch_autograd(kw):
    """Row a12 (restore_and_get_saliencies, trainers.py:703-732): d loss / d encoder_inputs from the oracle's manual
    chain (conv back-projection, un-im2row, un-reverse) against autograd through the independent torch model.  Padding
    samples are not inputs: the oracle defines their gradient as zero [BUILD-DEFINES]."""
    spec = tiny_spec(**kw)
    P = O.init_params(spec, seed=4)
    batch = make_batch(spec, B=5, T=11, L=6, seed=3, ragged=True)
    train = spec.ff_dropout > 0 or spec.rnn_dropout > 0
    _, cache = O.forward(P, spec, batch, train=train, seed=5)
    O.backward(P, cache)
    got = O.input_gradient(P, cache)
    Pt = {k: torch.tensor(v) for k, v in P.items()}
    X = torch.tensor(batch['encoder_inputs'], requires_grad=True)
    torch_model(Pt, spec, batch, collect_masks(cache, spec), x_leaf=X)['total'].backward()
    want = X.grad.numpy()
    lens = cache['lens']
    assert np.abs(got).max() > 0
    for b in range(got.shape[0]):
        np.testing.assert_allclose(got[b, :lens[b]], want[b, :lens[b]], atol=1e-10, rtol=1e-8)
        assert not got[b, lens[b]:].any()
