"""Beam search (`beam_width` > 1, `temperature`: manifest keys the net must accept, reference
ecog2txt/auxiliary/EFC/mocha-1_word_sequence.yaml:31, 82).  The rule lives in the absent `machine_learning` package, so it is
[BUILD-DEFINES] (oracle/seq2seq.py beam_decode: tf.contrib.seq2seq.BeamSearchDecoder's, length penalty 0).
CPU: properties of the oracle.  GPU: the HIP path (e2t_beam_step / e2t_beam_reorder through Seq2SeqEngine.beam_decode and
SequenceNetwork's assessment) against the oracle."""
import itertools

import numpy as np
import pytest

from oracle import seq2seq as O
from helpers import tiny_spec, make_batch


def _model(seed=1, scale=0.3, **kw):
    spec = tiny_spec(**kw)
    P = O.init_params(spec, seed=seed)
    rng = np.random.default_rng(seed)
    for k in P:
        P[k] = P[k] + scale * rng.standard_normal(P[k].shape)
    return spec, P


def test_oracle_beam_width_one_is_greedy_and_wider_beams_never_score_lower():
    spec, P = _model()
    batch = make_batch(spec, B=7, T=12, L=6, seed=2)
    g, _ = O.greedy_decode(P, spec, batch, max_len=6)
    b1, s1 = O.beam_decode(P, spec, batch, 1, max_len=6)
    np.testing.assert_array_equal(g, b1)
    prev = s1[:, 0]
    for W in (2, 3, 5):
        toks, sc = O.beam_decode(P, spec, batch, W, max_len=6)
        assert (np.diff(sc, axis=1) <= 1e-12).all()                  # beams are kept best first
        assert (sc[:, 0] >= prev - 1e-9).all()
        prev = sc[:, 0]


def test_oracle_beam_search_finds_the_exhaustive_optimum_when_the_beam_is_wide_enough():
    """With W >= V^(L-1) nothing is ever pruned: the result is the best of ALL token sequences (finished ones frozen)."""
    spec, P = _model(seed=3, vocab=4)
    batch = make_batch(spec, B=3, T=9, L=3, seed=5)
    L, V = 3, spec.vocab
    toks, sc = O.beam_decode(P, spec, batch, V ** (L - 1), max_len=L, temperature=0.7)
    # exhaustive: score every sequence by teacher forcing through the same decoder (a sequence ends at its first <EOS>)
    B = 3
    best = np.full(B, -np.inf); best_seq = [None] * B
    for seq in itertools.product(range(V), repeat=L):
        y = np.array(seq)
        if O.EOS_ID in seq:
            cut = seq.index(O.EOS_ID)
            if any(t != O.PAD_ID for t in seq[cut + 1:]):
                continue                                              # canonical form: pads behind <EOS>
        # log-probability of the sequence, token by token (teacher forced), with the temperature
        Yt = np.tile(y[None], (B, 1))
        lp = _sequence_logprob(P, spec, batch, Yt, 0.7)
        for b in range(B):
            if lp[b] > best[b] + 1e-12:
                best[b], best_seq[b] = lp[b], seq
    np.testing.assert_allclose(sc[:, 0], best, rtol=0, atol=1e-9)
    for b in range(B):
        assert tuple(toks[b]) == tuple(best_seq[b])


def _sequence_logprob(P, spec, batch, Y, temperature):
    """sum of log softmax(logits / T) over the tokens of Y up to and including its first <EOS> (tokens behind it must be pads)."""
    q = O._identity
    dummy = dict(batch)
    B, L = Y.shape
    dummy['decoder_targets'] = np.full((B, 1), O.EOS_ID, np.int64)
    dummy.pop('encoder_targets', None)
    _, cache = O.forward(P, spec, dummy, train=False)
    h, c = cache['h0'], cache['c0']
    Emb = P['seq2seq/decoder_embedding_%d_%d_0/weights' % (spec.vocab, spec.dec_embed)]
    Kx, Kh = O._split_kernel(P['seq2seq/decoder_rnn/cell_0/kernel'], spec.dec_embed)
    bias = P['seq2seq/decoder_rnn/cell_0/bias']
    pn = O.ff_names('decoder_projection', [spec.dec_rnn] + list(spec.dec_proj_hidden) + [spec.vocab])
    H = spec.dec_rnn
    u = np.full(B, O.EOS_ID, np.int64)
    lp = np.zeros(B); alive = np.ones(B, bool)
    for l in range(L):
        z = Emb[u] @ Kx + bias + h @ Kh
        i = O.sigmoid(z[:, :H]); j = np.tanh(z[:, H:2 * H]); f = O.sigmoid(z[:, 2 * H:3 * H] + spec.forget_bias); o = O.sigmoid(z[:, 3 * H:])
        c = f * c + i * j
        h = o * np.tanh(c)
        logits, _ = O.ff_fwd(P, pn, h, spec, q, False, 0, 0)
        x = logits / temperature
        x = x - x.max(-1, keepdims=True)
        logp = x - np.log(np.exp(x).sum(-1, keepdims=True))
        lp += np.where(alive, logp[np.arange(B), Y[:, l]], 0.0)
        alive = alive & (Y[:, l] != O.EOS_ID)
        u = Y[:, l]
    return lp


@pytest.mark.gpu
@pytest.mark.parametrize('name,B,W,temp', [('small_dropout', 19, 1, 1.0), ('small_dropout', 19, 3, 1.0), ('cat_aux_hidden_proj', 19, 4, 0.6),
                                          ('tiny_odd', 5, 16, 1.3), ('cfg2_widths', 24, 2, 1.0)])
def test_hip_beam_search_matches_the_oracle(name, B, W, temp):
    import torch
    from test_gpu_parity import build, SPECS
    T, L = 26, 6
    eng, ws, ospec, P, batch = build(SPECS[name], B, T, L, seed=12, ragged=True)
    hyp, score = eng.beam_decode(ws, W, temp, which='p')
    torch.cuda.synchronize()
    hyp, score = hyp.cpu().numpy(), score.cpu().numpy()
    want, wscore = O.beam_decode(P, ospec, batch, W, max_len=L, temperature=temp, emulate_bf16=True)
    # scores: fp32 log-softmax on the device against fp64 (the logits themselves differ by bf16 rounding flips: 3e-2)
    fin = np.isfinite(wscore)
    assert (np.isfinite(score) == fin).all()
    np.testing.assert_allclose(score[fin], wscore[fin], atol=6e-2 * L, rtol=0)
    # token sequences: identical wherever the oracle's decision margins are clear of that noise
    clear = (wscore[:, 0] - (wscore[:, 1] if W > 1 else -np.inf)) > 0.25 * L if W > 1 else np.ones(B, bool)
    if W == 1:
        g = eng.greedy_decode(ws, which='p').cpu().numpy()
        np.testing.assert_array_equal(hyp, g)                         # width 1 IS greedy decoding, bit for bit
    same = (hyp == want).all(axis=1)
    assert same[clear].all(), (hyp[clear & ~same][:3], want[clear & ~same][:3])
    assert same.mean() > 0.7
