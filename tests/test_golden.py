"""Golden vectors (tests/golden/seq2seq_small.npz, made by tests/golden/make_golden.py from the oracle):
 * CPU: the oracle still reproduces them bit-for-bit (pins the oracle against silent edits);
 * GPU: the HIP path reproduces the bf16-emulated vectors within the stated tolerances."""
import json
import os

import numpy as np
import pytest

from oracle import seq2seq as O

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, 'golden', 'seq2seq_small.npz'))
    spec_kw = json.loads(str(z['spec_json']))
    P = {k[2:]: z[k] for k in z.files if k.startswith('P/')}
    batch = {k[6:]: z[k] for k in z.files if k.startswith('batch/')}
    batch['subnet_id'] = '401'
    return z, spec_kw, P, batch


def test_oracle_reproduces_golden():
    z, spec_kw, P, batch = load()
    spec = O.NetSpec(**spec_kw)
    for mode, emu in (('exact', False), ('bf16', True)):
        losses, cache = O.forward(P, spec, batch, train=True, seed=31, emulate_bf16=emu)
        G = O.backward(P, cache)
        np.testing.assert_array_equal(z['%s/losses' % mode], [losses['decoder'], losses['aux'], losses['accuracy'], losses['total']])
        np.testing.assert_array_equal(z['%s/logits' % mode], cache['dec']['logits'])
        for k, v in G.items():
            np.testing.assert_array_equal(z['%s/G/%s' % (mode, k)], v)
        hyp, _ = O.greedy_decode(P, spec, batch, max_len=5, emulate_bf16=emu)
        np.testing.assert_array_equal(z['%s/greedy' % mode], hyp)


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    import torch
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    z, spec_kw, P, batch = load()
    eng = Seq2SeqEngine(NetSpec(**spec_kw), device='cuda:0', seed=31)
    eng.load_params(P)
    ws = eng.workspace('401', 6, 22, 5)
    eng.set_batch(ws, batch)
    eng.forward(ws, train=True)
    eng.backward(ws, train=True)
    torch.cuda.synchronize()
    got = eng.losses(ws)
    want = z['bf16/losses']
    assert abs(got['decoder'] - want[0]) <= 2e-4 * max(1, abs(want[0]))
    assert abs(got['aux'] - want[1]) <= 2e-4 * max(1, abs(want[1]))
    np.testing.assert_allclose(ws['proj']['out'].cpu().numpy().reshape(5, 6, -1), z['bf16/logits'], atol=3e-2, rtol=1e-2)
    Gd = eng.store.export_tf('g')
    from test_gpu_parity import check_grad
    for k in Gd:
        check_grad(k, Gd[k], z['bf16/G/' + k])
    hyp = eng.greedy_decode(ws, which='p').cpu().numpy()
    np.testing.assert_array_equal(hyp, z['bf16/greedy'])
    eng.adam_step('401')
    torch.cuda.synchronize()
    Pd = eng.store.export_tf('p')
    for k in Pd:
        assert np.abs(Pd[k] - z['adam/P/' + k]).max() < 2.5e-4, k     # |step| <= lr = 5e-4 per coordinate
