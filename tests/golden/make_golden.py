"""Generates tests/golden/seq2seq_small.npz from the oracle (seeded).  Committed so the vectors can be
regenerated; the reference itself has no golden vectors for this path.
    python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import seq2seq as O      # noqa: E402
from helpers import make_batch       # noqa: E402

SPEC = dict(channels={'401': 8}, decimation=4, enc_embed=12, enc_rnn=[8, 8], dec_embed=6, dec_rnn=16, vocab=17,
            aux_layer=1, aux_hidden=[10], aux_dim=3, ff_dropout=0.1, rnn_dropout=0.25)


def main():
    spec = O.NetSpec(**SPEC)
    P = O.init_params(spec, seed=21)
    rng = np.random.default_rng(22)
    for k in P:
        if P[k].ndim == 1:
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    batch = make_batch(spec, B=6, T=22, L=5, seed=23, ragged=True)
    out = {'spec_json': np.array(json.dumps(SPEC))}
    out.update({'P/' + k: v for k, v in P.items()})
    out.update({'batch/' + k: v for k, v in batch.items() if k != 'subnet_id'})
    for mode, emu in (('exact', False), ('bf16', True)):
        losses, cache = O.forward(P, spec, batch, train=True, seed=31, emulate_bf16=emu)
        G = O.backward(P, cache)
        out['%s/losses' % mode] = np.array([losses['decoder'], losses['aux'], losses['accuracy'], losses['total']])
        out['%s/logits' % mode] = cache['dec']['logits']
        out.update({'%s/G/%s' % (mode, k): v for k, v in G.items()})
        hyp, _ = O.greedy_decode(P, spec, batch, max_len=5, emulate_bf16=emu)
        out['%s/greedy' % mode] = hyp
    P2, st = O.adam_ema_step({k: v.copy() for k, v in P.items()}, G, {}, lr=5e-4)
    out.update({'adam/P/' + k: v for k, v in P2.items()})
    np.savez_compressed(os.path.join(HERE, 'seq2seq_small.npz'), **out)
    print('wrote', os.path.join(HERE, 'seq2seq_small.npz'))


if __name__ == '__main__':
    main()
