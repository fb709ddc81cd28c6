"""Shared synthetic-batch builders for the parity tests."""
import numpy as np

from oracle import seq2seq as O


def tiny_spec(**kw):
    d = dict(channels={401: 6}, decimation=3, enc_embed=5, enc_rnn=[4, 4], dec_embed=3, dec_rnn=8,
             vocab=11, aux_layer=1, aux_hidden=[6], aux_dim=2, ff_dropout=0.0, rnn_dropout=0.0)
    d.update(kw)
    return O.NetSpec(**d)


def make_batch(spec, B, T, L, seed=0, ragged=True, sid=None, categorical=False):
    rng = np.random.default_rng(seed)
    sid = sid if sid is not None else list(spec.channels)[0]
    C = spec.channels[sid]
    X = np.abs(rng.standard_normal((B, T, C))) + 0.05
    lens = rng.integers(max(1, T // 2), T + 1, size=B) if ragged else np.full(B, T)
    lens[0] = T
    if ragged and B > 2:
        lens[1] = 1
    for b in range(B):
        X[b, lens[b]:] = 0.0
    Y = np.zeros((B, L), np.int64)
    dl = rng.integers(min(2, L), L + 1, size=B) if ragged else np.full(B, L)
    dl[0] = L
    for b in range(B):
        n = dl[b]
        Y[b, :n - 1] = rng.integers(3, spec.vocab, size=n - 1)
        Y[b, n - 1] = O.EOS_ID
    batch = dict(subnet_id=sid, encoder_inputs=X, decoder_targets=Y)
    if spec.aux_layer is not None:
        if categorical:
            A = rng.integers(1, spec.aux_dim, size=(B, T))
            for b in range(B):
                A[b, lens[b]:] = 0
        else:
            A = rng.standard_normal((B, T, spec.aux_dim))
            A[np.abs(A) < 1e-3] = 0.5
            for b in range(B):
                A[b, lens[b]:] = 0.0
        batch['encoder_targets'] = A
    extra = []
    for hx in getattr(spec, 'aux_extra', []):
        if hx.get('dist', 'Gaussian') == 'categorical':
            A = rng.integers(1, hx['dim'], size=(B, T))
            for b in range(B):
                A[b, lens[b]:] = 0
        else:
            A = rng.standard_normal((B, T, hx['dim']))
            A[np.abs(A) < 1e-3] = 0.5
            for b in range(B):
                A[b, lens[b]:] = 0.0
        extra.append(A)
    if extra:
        batch['encoder_targets_extra'] = extra
    return batch
