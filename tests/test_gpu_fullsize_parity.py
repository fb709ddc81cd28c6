"""Full-size parity leg: the REAL graphs of BASELINE.json's configurations -- cfg2 (256 electrodes x
400 samples, 3 x biLSTM(400), decoder 800, V = 1806, B = 256: 8704-row GEMMs with split-K, S = 34 persistent sweeps, K = 3072
conv), cfg4 (H = 1024 x 4 layers, decoder 2048) and cfg5 (1024 electrodes x 2000 samples: S = 167 persistent sweeps, the
K = 12 288 one-pass front-end the engine picks for HBM-sized batches, fp32 inputs and bf16-staged inputs) -- compared tensor by
tensor with CPU checkers:

 * HIP path vs `oracle/torch_model.py` (the independent torch-CPU model, fp32, autograd) at the full batch.  The HIP path
   multiplies bf16 operands, so the tolerance is the measured bf16-vs-exact band, stated per quantity:
   losses 1e-2 relative, logits cosine >= 0.999, every gradient tensor cosine >= 0.995 and norm ratio within 3 %.
 * HIP path vs the bf16-emulating NumPy oracle (`oracle/seq2seq.py`, same rounding points) at B = 16 of the SAME cfg2
   graph, with the tolerances of tests/test_gpu_parity.py (losses 2e-4, gradients 5e-3 of the tensor's maximum).

Dropout is off in the torch legs (the torch model draws Bernoulli masks, not Philox); the oracle leg runs twice, the second time
with dropout ON: Philox4x32-10 on both sides, masks on 8704-row tensors (conv output, layer outputs, the 225-wide auxiliary
layer whose rows straddle two Philox blocks, decoder input / output) -- the same tolerances.
"""
import numpy as np
import pytest
import torch

import bench
from oracle import seq2seq as O

pytestmark = pytest.mark.gpu


def _ragged(batch, T, lo, seed=0):
    rng = np.random.default_rng(seed)
    B = batch['encoder_inputs'].shape[0]
    lens = rng.integers(lo, T + 1, size=B)
    lens[0] = T
    for b in range(B):
        batch['encoder_inputs'][b, lens[b]:] = 0
        if 'encoder_targets' in batch:
            batch['encoder_targets'][b, lens[b]:] = 0
    return lens


def _hip(kw, B, T, L, batch, P, train=False, prepare=None):
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    eng = Seq2SeqEngine(NetSpec(**kw), device='cuda:0', seed=5)
    eng.load_params(P)
    ws = eng.workspace(401, B, T, L)
    eng.set_batch(ws, batch)
    if prepare is not None:
        prepare(eng, ws)
    eng.forward(ws, train=train)
    eng.backward(ws, train=train)
    torch.cuda.synchronize()
    assert int(eng.sync_err[0].item()) == 0
    losses = eng.losses(ws)
    logits = ws['proj']['out'].float().cpu().numpy().reshape(L, B, -1)
    return eng, ws, losses, logits, eng.store.export_tf('g')


def _torch_reference(ospec, batch, P, dtype=torch.float32):
    from oracle import torch_model as TM
    Pt = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in P.items()}
    b = dict(batch)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    b['encoder_inputs'] = np.asarray(batch['encoder_inputs'], dtype=npdt)
    if 'encoder_targets' in b:
        b['encoder_targets'] = np.asarray(batch['encoder_targets'], dtype=npdt)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    out = TM.torch_model(Pt, ospec, b, {})
    out['total'].backward()
    G = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, npdt)) for k, v in Pt.items()}
    losses = {k: float(v.detach()) for k, v in out.items() if k != 'logits'}
    return losses, out['logits'].detach().numpy(), G


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _biases_off_zero(P, seed):
    """Random biases (the initialiser leaves them at zero: a transposed or mis-sliced bias gradient would go unseen)."""
    rng = np.random.default_rng(seed)
    for k in P:
        if P[k].ndim == 1:
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    return P


def _stage_bf16(eng, ws):
    """The batch as bf16-staged inputs (Seq2SeqEngine.pack_inputs / load_packed_batch: SURVEY.md 8 d4 'bf16 in'), rows in a
    shuffled partition with padding utterances in between, as a fit assembles them."""
    B = ws['B']
    rng = np.random.default_rng(5)
    n = B + 9
    where = rng.permutation(n)[:B]                       # utterance b of the batch is row where[b] of the partition
    Xp = torch.zeros(n, ws['T'], ws['C'], device='cuda')
    Xp[torch.from_numpy(where).cuda()] = ws['X']
    pk = eng.pack_inputs(ws['sid'], Xp)
    ws['X'].fill_(float('nan'))                           # the step must not touch x any more
    eng.load_packed_batch(ws, pk, torch.from_numpy(where.astype(np.int32)).cuda())


@pytest.mark.parametrize('name,B', [('cfg2', 256), ('cfg4', 64), ('cfg5', 64), ('cfg5_bf16_staged', 64)])
def test_full_graph_against_torch_cpu(name, B):
    staged = name.endswith('_bf16_staged')
    name = name.replace('_bf16_staged', '')
    kw, _, T, L = bench.CONFIGS[name]
    from ecog2txt_amd.engine import NetSpec
    ospec = O.NetSpec(**NetSpec(**kw).as_dict())
    P = _biases_off_zero(O.init_params(ospec, seed=3), 4)
    batch = bench.synth_batch(kw, B, T, L, seed=7)
    _ragged(batch, T, 240, seed=1)
    eng, ws, losses, logits, G = _hip(kw, B, T, L, batch, P, prepare=_stage_bf16 if staged else None)
    if staged:
        assert ws['packed'] and torch.isnan(ws['X']).all()
    elif name == 'cfg5':
        # the engine took the one-pass front-end (e2t_conv_fwd_fused), as it does at B = 256: the batch is HBM-sized (524 MB)
        assert B * T * kw['channels'][401] * 4 >= (1 << 28) and eng.fused_conv == 'auto' and ws['A_stale'] is False
    want, wlogits, WG = _torch_reference(ospec, batch, P)
    # losses: 1e-2 relative (bf16 operands against fp32)
    for k in ('decoder', 'aux'):
        if k in want:
            assert abs(losses[k] - want[k]) <= 1e-2 * max(1.0, abs(want[k])), (k, losses, want)
    # logits of the valid target positions
    Y = np.asarray(batch['decoder_targets'])
    valid = (Y != 0).T                                              # [L, B]
    assert _cos(logits[valid], wlogits.reshape(L, B, -1)[valid]) >= 0.999
    worst = []
    for k in sorted(WG):
        g, w = np.asarray(G[k], np.float64), np.asarray(WG[k], np.float64)
        assert g.shape == w.shape, (k, g.shape, w.shape)
        nw = np.linalg.norm(w)
        if nw == 0.0:
            assert np.abs(g).max() < 1e-6, k
            continue
        c, ratio = _cos(g, w), float(np.linalg.norm(g) / nw)
        worst.append((c, ratio, k))
        assert c >= 0.995, (k, c, ratio)
        assert abs(ratio - 1.0) <= 3e-2, (k, c, ratio)
    worst.sort()
    print('\n%s B=%d: losses hip %s torch %s; worst gradient cosines: %s' % (
        name, B, {k: round(v, 5) for k, v in losses.items() if k in want}, {k: round(v, 5) for k, v in want.items()},
        ['%s cos %.5f ratio %.4f' % (k, c, r) for c, r, k in worst[:4]]))


@pytest.mark.parametrize('train', [False, True], ids=['dropout_off', 'dropout_on'])
def test_cfg2_graph_against_bf16_emulating_oracle(train):
    """The same cfg2 graph (3 x 400 bidirectional, decoder 800, V = 1806, T = 400 -> 34 steps, K = 3072 conv) at B = 16 against the
    NumPy oracle with the device's rounding points: the tight tolerances of test_gpu_parity.py.  train=True: FF dropout 0.1 and
    RNN dropout 0.5 on, Philox masks on both sides."""
    from test_gpu_parity import check_grad, relu_class, LOSS_RTOL
    from ecog2txt_amd.engine import NetSpec
    kw, _, T, L = bench.CONFIGS['cfg2']
    B = 16
    ospec = O.NetSpec(**NetSpec(**kw).as_dict())
    P = _biases_off_zero(O.init_params(ospec, seed=5), 6)
    batch = bench.synth_batch(kw, B, T, L, seed=9)
    _ragged(batch, T, 200, seed=2)
    eng, ws, losses, logits, G = _hip(kw, B, T, L, batch, P, train=train)
    want, cache = O.forward(P, ospec, batch, train=train, seed=5, emulate_bf16=True)
    for k in ('decoder', 'aux'):
        assert abs(losses[k] - want[k]) <= LOSS_RTOL * max(1.0, abs(want[k])), (k, losses, want)
    np.testing.assert_array_equal(ws['lens'].cpu().numpy(), cache['lens'])
    np.testing.assert_allclose(logits, cache['dec']['logits'], atol=3e-2, rtol=1e-2)
    WG = O.backward(P, cache)
    if train:
        # the oracle's own distance between its bf16 mode and the exact fp64 spec: the yardstick of the dropout-on leg
        _, cache_x = O.forward(P, ospec, batch, train=True, seed=5, emulate_bf16=False)
        XG = O.backward(P, cache_x)
    rl2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-12))
    for k in sorted(WG):
        # 54 400 conv activations here (16 utterances x 34 steps x 100 units) against ~2 000 in the small cases: a few units sit
        # on the ReLU knife edge, each moving one column (1 %) of the conv weight gradient
        # ... and 1.7 M input projections per layer are rounded to bf16 (544 per bias element): two or three entries of a 1600-entry
        # bias gradient beyond 5e-3 (measured 6.3e-3 at most)
        if not train:
            check_grad(k, G[k], WG[k], relu_outliers=4e-2, flip_outliers=3e-3)
            continue
        # dropout on (scripts/diag_fullsize_dropout.py): every kept activation carries 1 / keep = 2x, so a 1-ulp bf16 flip or a
        # flipped ReLU unit of the 225-wide auxiliary layer weighs twice as much in everything below the tapped layer, and the
        # share of entries beyond a fixed 5e-3 grows with it.  No outlier allowance is made for that (round 5): THE assertion is
        # relative to the oracle's own band -- the HIP path must be CLOSER to the bf16-emulating oracle than 0.6 of that oracle's
        # distance from the exact fp64 spec (measured: 0.2 .. 0.46) -- plus the hard ceilings of test_gpu_parity.py on the single
        # worst entry (2e-2 of the tensor's maximum; 5e-2 behind a ReLU mask of the tensor's own layer)
        band = rl2(XG[k], WG[k])
        assert rl2(G[k], WG[k]) <= max(0.6 * band, 2e-3), (k, rl2(G[k], WG[k]), band)
        worst = float(np.abs(G[k] - WG[k]).max() / (np.abs(WG[k]).max() + 1e-12))
        assert worst < (5e-2 if relu_class(k) else 2e-2), (k, worst)
