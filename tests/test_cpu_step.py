"""oracle/cpu_step.cpp (the C++ / OpenMP fp32 CPU train step timed as `cpu_baseline`) against the NumPy
oracle: losses, every gradient, and the parameters after two Adam + EMA steps, on ragged batches with dropout off."""
import copy
import shutil

import numpy as np
import pytest

from oracle import seq2seq as O
from oracle import cpu_step as CS


def _batch(rng, spec, sid, B, T, L):
    C = spec.channels[sid]
    X = rng.standard_normal((B, T, C)).astype(np.float32)
    lens = rng.integers(max(1, T // 2), T + 1, size=B)
    lens[0] = T
    for b in range(B):
        X[b, lens[b]:] = 0
    X[np.abs(X) < 1e-6] = 1e-3
    Y = np.zeros((B, L), np.int32)
    for b in range(B):
        n = rng.integers(1, L)
        Y[b, :n] = rng.integers(3, spec.vocab, size=n)
        Y[b, n] = 1
    A = rng.standard_normal((B, T, spec.aux_dim)).astype(np.float32)
    for b in range(B):
        A[b, lens[b]:] = 0
    return dict(subnet_id=sid, encoder_inputs=X, decoder_targets=Y, encoder_targets=A)


pytestmark = pytest.mark.skipif(shutil.which('g++') is None, reason='the CPU step is built with g++ on the box that runs it')


@pytest.mark.parametrize('aux_layer,enc', [(1, [6, 10]), (0, [8]), (2, [4, 6, 8])])
def test_cpu_step_matches_the_numpy_oracle(aux_layer, enc):
    rng = np.random.default_rng(5 + aux_layer)
    spec = O.NetSpec(channels={7: 9}, decimation=3, enc_embed=7, enc_rnn=enc, dec_embed=5, dec_rnn=2 * enc[-1], vocab=23,
                     aux_layer=aux_layer, aux_hidden=[11], aux_dim=4, ff_dropout=0.0, rnn_dropout=0.0, aux_scale=0.7, dec_scale=1.3)
    B, T, L = 37, 20, 6                       # B not a multiple of the GEMM's row block, ragged T
    P = O.init_params(spec, seed=3)
    for k in P:                               # non-zero biases
        if k.endswith('bias') or k.endswith('biases'):
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    batch = _batch(rng, spec, 7, B, T, L)
    cpu = CS.CpuStep(spec, 7, B, T, L)
    cpu.load_params(P)
    Pn = copy.deepcopy(P)
    state = {}
    for it in range(2):
        losses, cache = O.forward(Pn, spec, batch, train=False)
        G = O.backward(Pn, cache)
        got = cpu.fwd_bwd(batch, train=False)
        assert abs(got['decoder'] - losses['decoder']) < 2e-5 * max(1, abs(losses['decoder']))
        assert abs(got['aux'] - losses['aux']) < 2e-5 * max(1, abs(losses['aux']))
        assert abs(got['accuracy'] - losses['accuracy']) < 1e-6
        Gc = cpu.grads()
        for k in cpu.names:
            scale = max(np.abs(G[k]).max(), 1e-6)
            np.testing.assert_allclose(Gc[k], G[k], rtol=0, atol=3e-5 * scale, err_msg=k)
        Pn, state = O.adam_ema_step(Pn, G, state)
        cpu.adam()
    Pc = cpu.params()
    for k in cpu.names:
        np.testing.assert_allclose(Pc[k], Pn[k], rtol=0, atol=2e-5, err_msg=k)
    ema = cpu._unflat(cpu._view('e2t_cpu_ema'))
    for k in cpu.names:
        np.testing.assert_allclose(ema[k], state['ema'][k], rtol=0, atol=2e-5, err_msg=k)
    cpu.close()
