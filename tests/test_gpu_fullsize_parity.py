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


def _rl2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-12))


# Tolerances of the oracle legs.
#  * `off` = (relu_outliers, flip_outliers, l2_scale) handed to test_gpu_parity.check_grad with dropout OFF: the share of a tensor's
#    entries that may lie beyond 5e-3 of its maximum (ReLU class / everything else) and the scale of the relative-L2 ceilings
#    (2e-2 / 1e-2).  cfg2 / cfg5 (3 x 400): the small-case tolerances.
#  * `band`: how close the HIP path must be to the bf16-emulating oracle, in units of that oracle's OWN distance from the exact fp64
#    spec on the same tensor (relative L2; "the band").  With dropout on every kept activation carries 1 / keep = 2x, a 1-ulp bf16
#    flip or a flipped ReLU unit of the 225-wide auxiliary layer weighs twice as much in everything below the tapped layer
#    (scripts/diag_fullsize_dropout.py), and fixed element-wise thresholds say little: the band-relative assertions scale with the
#    problem.  A flip -- fp32 against fp64 accumulation, v_exp / v_rcp against libm -- perturbs everything downstream and begets
#    further flips, so the two implementations decorrelate with depth and width: measured (round 6) 0.43 .. 0.59 of the band at
#    3 x 400 (cfg2, cfg5), 0.63 .. 0.74 at 4 x 1024 (cfg4: K = 3072 gate sums), up to 0.78 on cfg3's third participant, whose band is
#    itself 3.3e-2; two INDEPENDENT bf16 implementations would sit at 1.41.
#  * `tail`: at most 1 % of a tensor's entries may be further from the oracle than `tail` x the RMS of the oracle's own rounding
#    error on that tensor, and no entry further than the fixed ceilings of test_gpu_parity.py (2e-2 of the tensor's maximum; 5e-2
#    behind a ReLU mask of the tensor's own layer) or 3 bands, whichever is larger.
#  * what keeps the band-relative assertions honest is the last one: the HIP path is no further from the EXACT spec than decorrelated
#    rounding of the measured size explains, d(HIP, exact) <= 1.08 sqrt(band^2 + d(HIP, oracle)^2) -- an error that is systematic
#    (a kernel bug) rather than decorrelated rounding adds up with the oracle's own and fails that (measured: 0.92 .. 1.23 band, all
#    within 1.00 .. 1.02 of the root sum).
TOL = {
    'default': dict(band=0.7, tail=4.0, off=(4e-2, 3e-3, 1.0)),
    'cfg4': dict(band=1.0, tail=4.0, off=(0.2, 6e-3, 1.5)),
    'cfg3': dict(band=1.0, tail=4.0, off=(4e-2, 3e-3, 1.0)),
}


def _check_against_oracle(tag, losses, G, want, WG, XG, train, report, tol=None):
    """Losses and every gradient tensor of one forward / backward pass of the HIP path against the bf16-emulating oracle (WG).  XG =
    the oracle's gradients in its EXACT fp64 mode: the yardstick of the relative criteria.  Every tensor is looked at before
    anything is raised (the message lists all offenders)."""
    from test_gpu_parity import check_grad, relu_class, LOSS_RTOL, GRAD_TOL
    tol = tol or TOL['default']
    for k in ('decoder', 'aux'):
        if k in want:
            assert abs(losses[k] - want[k]) <= LOSS_RTOL * max(1.0, abs(want[k])), (tag, k, losses, want)
    failed = []
    for k in sorted(WG):
        g, w, x = np.asarray(G[k], np.float64), np.asarray(WG[k], np.float64), np.asarray(XG[k], np.float64)
        scale = np.abs(w).max() + 1e-12
        err = np.abs(g - w) / scale
        worst = float(err.max())
        band = _rl2(x, w)                            # bf16-emulating oracle <-> exact spec
        d_or, d_ex = _rl2(g, w), _rl2(g, x)
        rms_band = float(np.sqrt(np.mean((x - w) ** 2))) / scale
        tail = float((err > tol['tail'] * max(rms_band, 1e-4)).mean())
        report.append((worst, d_or, float((err > GRAD_TOL).mean()), d_or / (band + 1e-30), d_ex / (band + 1e-30), tail, tag, k))
        try:
            if not train:
                # dropout off: the fixed element-wise tolerances (worst entry, share beyond 5e-3, relative L2)
                check_grad(k, G[k], WG[k], relu_outliers=tol['off'][0], flip_outliers=tol['off'][1], l2_scale=tol['off'][2])
            assert d_or <= max(tol['band'] * band, 2e-3), (tag, k, 'relative to the band', d_or, band)
            assert tail <= max(1e-2, 2.0 / err.size), (tag, k, 'share of entries beyond %.1f RMS of the band' % tol['tail'], tail)
            assert worst < max(5e-2 if relu_class(k) else 2e-2, 3.0 * band), (tag, k, 'worst entry', worst, band)
            assert d_ex <= max(1.08 * float(np.hypot(band, d_or)), 2e-3), (tag, k, 'distance from the exact fp64 spec', d_ex, band, d_or)
        except AssertionError as e:
            failed.append(str(e).splitlines()[0][:300])
    assert not failed, '%d tensor(s) outside the tolerances:\n  ' % len(failed) + '\n  '.join(failed)


def _print_worst(title, report, n=5):
    report.sort(reverse=True)
    print('\n%s -- worst tensors (max error / max |g|, relative L2, share of entries beyond 5e-3, distance to the bf16-emulating '
          'oracle and to the exact fp64 spec in units of that oracle\'s own distance from the spec, share of entries beyond 4 RMS of it):' % title)
    for worst, rel, share, r_or, r_ex, tail, tag, k in report[:n]:
        print('  %-8s %-66s %.2e  %.2e  %.5f  %.2f  %.2f  %.5f' % (tag, k, worst, rel, share, r_or, r_ex, tail))
    if report:
        print('  over all %d tensors: distance to the oracle <= %.2f band, to the exact spec <= %.2f band, share beyond 4 RMS <= %.5f' % (
            len(report), max(r[3] for r in report), max(r[4] for r in report), max(r[5] for r in report)))


# the REAL graph of every BASELINE configuration against the oracle with the device's rounding points:
#   cfg2  3 x 400 bidirectional, decoder 800, V = 1806, T = 400 -> 34 sweeps, K = 3072 conv                           B = 16
#   cfg4  4 x lstm_big (H = 1024), 256 x 256 K-major products, k_colsum_bf16 bias rows, launch-per-step decoder 2048   B = 16
#   cfg5  1024 electrodes x 2000 samples: S = 167 persistent sweeps, K = 12 288, the ONE-PASS front-end the engine picks at
#         B = 256 (forced here: fused_conv='1'), and the bf16-staged input form                                        B = 8
ORACLE_LEGS = {
    'cfg2': ('cfg2', 16, 200, None, None),
    'cfg4': ('cfg4', 16, 200, None, None),
    'cfg5_one_pass_frontend': ('cfg5', 8, 1200, {'fused_conv': '1'}, None),
    'cfg5_bf16_staged': ('cfg5', 8, 1200, None, _stage_bf16),
}


@pytest.mark.parametrize('train', [False, True], ids=['dropout_off', 'dropout_on'])
@pytest.mark.parametrize('leg', list(ORACLE_LEGS))
def test_real_graph_against_bf16_emulating_oracle(leg, train):
    """The graphs of BASELINE.json's configurations at a batch the NumPy oracle finishes in a minute, against that oracle with
    the device's rounding points: the tight tolerances of test_gpu_parity.py (losses 2e-4, gradients 5e-3 of the tensor's
    maximum, lengths equal, logits 3e-2).  train=True: FF dropout 0.1 and RNN dropout 0.5 on, Philox masks on both sides.
    Kernel shapes: trainers.py:527-541; sizes: mocha-1_word_sequence.yaml:57-61."""
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    name, B, lo, options, prepare = ORACLE_LEGS[leg]
    kw, _, T, L = bench.CONFIGS[name]
    ospec = O.NetSpec(**NetSpec(**kw).as_dict())
    P = _biases_off_zero(O.init_params(ospec, seed=5), 6)
    batch = bench.synth_batch(kw, B, T, L, seed=9)
    _ragged(batch, T, lo, seed=2)
    eng = Seq2SeqEngine(NetSpec(**kw), device='cuda:0', seed=5, options=options)
    eng.load_params(P)
    ws = eng.workspace(401, B, T, L)
    eng.set_batch(ws, batch)
    if prepare is not None:
        prepare(eng, ws)
    eng.forward(ws, train=train)
    if leg == 'cfg5_one_pass_frontend':
        assert bool(ws['A_stale']) is (not train) and not ws.get('packed')    # e2t_conv_fwd_fused ran (inference leaves no im2row copy)
    if prepare is not None:
        assert ws['packed'] and torch.isnan(ws['X']).all()
    eng.backward(ws, train=train)
    torch.cuda.synchronize()
    assert int(eng.sync_err[0].item()) == 0
    losses = eng.losses(ws)
    logits = ws['proj']['out'].float().cpu().numpy().reshape(L, B, -1)
    G = eng.store.export_tf('g')
    want, cache = O.forward(P, ospec, batch, train=train, seed=5, emulate_bf16=True)
    np.testing.assert_array_equal(ws['lens'].cpu().numpy(), cache['lens'])
    np.testing.assert_allclose(logits, cache['dec']['logits'], atol=3e-2, rtol=1e-2)
    WG = O.backward(P, cache)
    # the oracle's own distance between its bf16 mode and the exact fp64 spec: the yardstick of the relative criteria
    _, cache_x = O.forward(P, ospec, batch, train=train, seed=5, emulate_bf16=False)
    XG = O.backward(P, cache_x)
    report = []
    try:
        _check_against_oracle(leg, losses, G, want, WG, XG, train, report, tol=TOL.get(name, TOL['default']))
    finally:
        _print_worst('%s B=%d dropout %s: losses hip %s oracle %s' % (
            leg, B, 'on' if train else 'off', {k: round(v, 6) for k, v in losses.items() if k in want},
            {k: round(float(v), 6) for k, v in want.items() if k in ('decoder', 'aux')}), report, n=8)


def test_cfg3_round_robin_round_against_bf16_emulating_oracle():
    """cfg3's single-GPU half at FULL widths (BASELINE.json configs[2]; grids 16x16 / 16x16 / 8x16 / 16x16:
    mochastar_word_sequence.yaml:57-59, 150-152, 243-245, 336-338; per-subject front-ends trainers.py:801-818): one round-robin
    round over the four participants at B = 16 with dropout on and Adam + EMA between the steps.  Every step: losses and all
    gradients (the participant's own front-end included) against the bf16-emulating oracle evaluated at the ORACLE's trajectory;
    afterwards parameters and EMA shadows of the round.  Then one more round replayed from the captured graphs (one per
    participant) -- the trajectory still follows."""
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec
    from test_gpu_parity import relu_class
    kw, _, T, L = bench.CONFIGS['cfg3']
    B, lr = 16, 5e-4
    spec = NetSpec(**kw)
    ospec = O.NetSpec(**spec.as_dict())
    P = _biases_off_zero(O.init_params(ospec, seed=5), 6)
    eng = Seq2SeqEngine(spec, device='cuda:0', seed=11, lr=lr)
    eng.load_params(P)
    batches, wss = {}, {}
    for i, sid in enumerate(kw['channels']):
        kw1 = dict(kw, channels={sid: kw['channels'][sid]})
        batches[sid] = bench.synth_batch(kw1, B, T, L, seed=20 + i)
        _ragged(batches[sid], T, 200, seed=30 + i)
        wss[sid] = eng.workspace(sid, B, T, L)
        eng.set_batch(wss[sid], batches[sid])
    Po = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in P.items()}     # the device's fp32 masters
    state, step, report = {}, 0, []
    first_g = {}                 # the oracle's gradient of each participant's front-end at its (one) step of the first round
    for sid in kw['channels']:
        ws = wss[sid]
        # (exactly train_step(use_graph=False), with the gradient read out in between)
        eng.forward(ws, train=True)
        eng.backward(ws, train=True)
        torch.cuda.synchronize()
        assert int(eng.sync_err[0].item()) == 0
        losses, G = eng.losses(ws), eng.store.export_tf('g')
        # the oracle at the DEVICE's current masters: each step's gradient comparison stands on its own (the trajectory legs
        # below carry the accumulated difference)
        Pd = {k: np.asarray(v, np.float64) for k, v in eng.store.export_tf('p').items()}
        want, cache = O.forward(Pd, ospec, batches[sid], train=True, seed=11 + step, emulate_bf16=True)
        WG = O.backward(Pd, cache)
        _, cache_x = O.forward(Pd, ospec, batches[sid], train=True, seed=11 + step, emulate_bf16=False)
        XG = O.backward(Pd, cache_x)
        own = O.conv_name(ospec, sid) + '/weights'      # (the oracle returns the stepping participant's front-end only)
        assert own in WG and np.abs(WG[own]).max() > 0 and sum('subnet_' in k for k in WG) == 2
        try:
            _check_against_oracle('sid %s' % sid, losses, G, want, WG, XG, True, report, tol=TOL['cfg3'])
        except AssertionError:
            _print_worst('cfg3, participant %s' % sid, report, n=8)
            raise
        eng.adam_step(sid)
        # the oracle's own trajectory
        _, c2 = O.forward(Po, ospec, batches[sid], train=True, seed=11 + step, emulate_bf16=True)
        Go = O.backward(Po, c2)
        first_g.update({k: Go[k] for k in Go if 'subnet_' in k})
        Po, state = O.adam_ema_step(Po, Go, state, lr=lr)
        step += 1
    _print_worst('cfg3 B=%d, one round over 4 participants, dropout on' % B, report)

    def follows(nsteps):
        torch.cuda.synchronize()
        assert int(eng.sync_err[0].item()) == 0 and int(eng.step_t.item()) == nsteps
        Pd, Ed = eng.store.export_tf('p'), eng.store.export_tf('ema')
        for k in Po:
            err = np.abs(Pd[k] - Po[k])
            relu = relu_class(k)
            # (the bands of tests/test_gpu_decode_fullsize.py::test_cfg2_three_adam_ema_steps_follow_the_oracle: Adam normalises every
            #  coordinate's step to ~lr; a coordinate whose gradient is within round-off of zero may step the other way)
            assert err.max() <= 2.0 * nsteps * lr * 1.01, (k, float(err.max()))
            # EMA shadows (decay 0.99): a coordinate that steps the other way at EVERY one of n steps is 2 lr i off after step i, its
            # shadow 0.01 lr n (n + 1) after n (1e-4 at n = 4, 3.6e-4 at n = 8): asserted at 0.6 of that (measured 1.4e-4 at n = 8)
            assert np.abs(Ed[k] - state['ema'][k]).max() < max(1e-4, 0.6 * 0.01 * lr * nsteps * (nsteps + 1)), k
            if 'subnet_' in k:
                # a participant's front-end takes ONE step per round, and that first step is c lr g / (|g| + eps') with c <= 1 (TF1 form:
                # at global step 3 a fresh tensor's step is 0.64 lr, eps' = 3e-7) -- nearly lr sign(g) whatever |g| is: a coordinate
                # whose gradient lies within the gradient error of zero may go either way (measured: 1.8 % of a 307 200-entry kernel
                # differ by more than a third of a step; the gradient legs above allow single entries 3e-2 of the maximum off, and
                # this leg compares TRAJECTORIES, whose shared body has drifted by then).  The statement is made where the sign is
                # decided -- |g| >= 0.15 of the tensor's maximum: there every coordinate took the oracle's step -- plus a ceiling on
                # the share of all coordinates that went the other way
                clear = np.abs(first_g[k]) >= 0.15 * np.abs(first_g[k]).max()
                assert clear.sum() >= min(1000, err.size // 10) and err[clear].max() < 0.35 * lr * (nsteps // 4), (k, int(clear.sum()), float(err[clear].max()))
                assert (err > 0.35 * lr * (nsteps // 4)).mean() < 5e-2, (k, float((err > 0.35 * lr).mean()))
                continue
            nflip = int((err > nsteps * lr * 0.35).sum())          # (a third of the way the tensor can have moved by then)
            assert nflip <= max(5, (2e-2 if relu else 5e-3) * err.size), (k, float(err.max()), nflip, err.size)
            disp = Po[k] - np.asarray(P[k], np.float32)
            if np.linalg.norm(disp) > 0:
                assert np.linalg.norm(Pd[k] - Po[k]) / np.linalg.norm(disp) < (0.17 if relu else 0.12), k
    follows(4)
    # a participant's front-end moved on its own step only: three of the four steps left it alone -- it is one Adam step from P
    for sid in kw['channels']:
        k = O.conv_name(ospec, sid) + '/weights'
        d = np.abs(eng.store.export_tf('p')[k] - np.asarray(P[k], np.float32))
        assert 0 < d.max() <= lr * 1.01, (sid, float(d.max()))
    # second round: captured graphs, one per participant
    for sid in kw['channels']:
        eng.train_step(wss[sid], use_graph=True)
        _, c2 = O.forward(Po, ospec, batches[sid], train=True, seed=11 + step, emulate_bf16=True)
        Po, state = O.adam_ema_step(Po, O.backward(Po, c2), state, lr=lr)
        step += 1
    follows(8)
