#!/usr/bin/env python
"""Headline benchmark: train-step utterances/s of the ECoG->text seq2seq hot path.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): one subject, 256-electrode
grid, T = 400 samples (decimation 12 -> 34 encoder steps), B = 256 utterances per GPU,
conv 100 -> 3 x biLSTM(400) -> aux head 225 -> 13 -> LSTM decoder 800 -> 1806 words,
bf16 operands / fp32 accumulate, dropout on, Adam + EMA.  A "step" is one full
optimisation step (forward, losses, backward, [all-reduce], Adam+EMA, operand re-pack)
on one batch of synthetic ECoG already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); weak scaling:
every rank steps its own B utterances, gradients are all-reduced every step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (spec kwargs, B, T, L)
    'cfg2': (dict(channels={401: 256}, decimation=12, enc_embed=100, enc_rnn=[400, 400, 400], dec_embed=150, dec_rnn=800,
                  vocab=1806, aux_layer=1, aux_hidden=[225], aux_dim=13, ff_dropout=0.1, rnn_dropout=0.5), 256, 400, 10),
    'cfg4': (dict(channels={401: 256}, decimation=12, enc_embed=100, enc_rnn=[1024] * 4, dec_embed=150, dec_rnn=2048,
                  vocab=1806, aux_layer=1, aux_hidden=[225], aux_dim=13, ff_dropout=0.1, rnn_dropout=0.5), 256, 400, 10),
    'cfg5': (dict(channels={401: 1024}, decimation=12, enc_embed=100, enc_rnn=[400, 400, 400], dec_embed=150, dec_rnn=800,
                  vocab=1806, aux_layer=1, aux_hidden=[225], aux_dim=13, ff_dropout=0.1, rnn_dropout=0.5), 256, 2000, 10),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak


def synth_batch(spec_kw, B, T, L, seed):
    """|N(0,1)| z-scored 'high-gamma' plus a sentence-dependent low-rank
    signal; 50 sentences of 3..L-1 words + <EOS>; 13-dim N(0,1) auxiliary targets."""
    rng = np.random.default_rng(seed)
    C = list(spec_kw['channels'].values())[0]
    V, K = spec_kw['vocab'], spec_kw['aux_dim']
    nsent = 50
    sent_len = rng.integers(3, L, size=nsent)
    sents = [rng.integers(3, V, size=n) for n in sent_len]
    basis = rng.standard_normal((nsent, 4, C)).astype(np.float32)
    which = rng.integers(0, nsent, size=B)
    X = np.abs(rng.standard_normal((B, T, C), dtype=np.float32))
    X = (X - X.mean()) / X.std()
    tt = np.linspace(0, 1, T, dtype=np.float32)[:, None]
    for b in range(B):
        s = which[b]
        X[b] += 0.5 * (np.sin(2 * np.pi * (1 + s % 5) * tt) * basis[s, 0] + tt * basis[s, 1])
    X[X == 0] = 1e-3
    Y = np.zeros((B, L), np.int32)
    for b in range(B):
        w = sents[which[b]]
        Y[b, :len(w)] = w
        Y[b, len(w)] = 1
    A = rng.standard_normal((B, T, K), dtype=np.float32)
    return dict(subnet_id=list(spec_kw['channels'])[0], encoder_inputs=X, decoder_targets=Y, encoder_targets=A)


def recurrent_flops_fwd(spec_kw, S, L):
    """recurrent-GEMM flops per utterance, forward."""
    f = sum(2 * S * 2 * H * 4 * H for H in spec_kw['enc_rnn'])
    Hd = spec_kw['dec_rnn']
    return f + L * 2 * Hd * 4 * Hd


def cpu_baseline(spec_kw, T, L, budget_s=20.0):
    """Oracle (NumPy fp64 restatement, kind 'port') timed on the host cores on a bounded sample."""
    from oracle import seq2seq as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    ospec = O.NetSpec(**spec_kw)
    P = O.init_params(ospec, seed=0)
    Bs = 32
    batch = synth_batch(spec_kw, Bs, T, L, seed=1)
    state = {}
    t0 = time.perf_counter()
    n = 0
    while True:
        _, cache = O.forward(P, ospec, batch, train=True, seed=n)
        G = O.backward(P, cache)
        P, state = O.adam_ema_step(P, G, state)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 3:
            break
    return dict(value=round(n * Bs / el, 3), unit='utterances/s', cores=int(threads), kind='port',
                sample='%d train steps of B=%d utterances (T=%d, same architecture) with the NumPy fp64 oracle, %.1f s'
                       % (n, Bs, T, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg2', choices=list(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU (default: the config\'s 256)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, ceil_div
    from ecog2txt_amd.parallel import GradSync, broadcast_flat

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    spec_kw, B, T, L = CONFIGS[args.config]
    B = args.batch or B
    spec = NetSpec(**spec_kw)
    eng = Seq2SeqEngine(spec, device='cuda:%d' % local_rank, seed=1234 + rank)
    eng.init_params(seed=0)
    broadcast_flat([eng.store.p, eng.store.ema])
    eng.pack('p')
    sid = list(spec.channels)[0]
    ws = eng.workspace(sid, B, T, L)
    eng.set_batch(ws, synth_batch(spec_kw, B, T, L, seed=100 + rank))
    sync = GradSync(eng.store.g) if world > 1 else None
    torch.cuda.synchronize()

    def step():
        eng.train_step(ws, use_graph=not args.no_graph, sync=sync)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    losses = eng.losses(ws)
    assert np.isfinite(losses['total']), losses

    # ---- roofline of the dominant kernel (rocprofv3: k_gemm_nt, 49% of the step's kernel time; its largest instance is
    #      the input projection of an encoder layer, Gx[S*B][8H] = Ydrop[S*B][2H] . Wx^T + b, 256x256 tiles) ----
    # Measured live: `reps` launches captured in a hipGraph on the bench stream, bracketed by HIP events on THAT
    # stream.  Algorithmic flops per launch = 2*M*N*K; algorithmic HBM bytes = A + B (bf16) + C (fp32).
    roof = None
    extra = {}
    if rank == 0:
        import ctypes as C
        from ecog2txt_amd.hip_lib import lib
        S = ceil_div(T, spec.decimation)
        lay, lw = eng.enc[1], ws['enc'][1]
        x = ws['enc'][0]['Ydrop'].data_ptr()

        def time_graph(fn, reps):
            fn(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    fn()
            gr.replay(); torch.cuda.synchronize()
            stream = torch.cuda.current_stream()          # graph replays are enqueued on this stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                gr.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (5 * reps)

        M, N, K = S * B, lay.N4, lay.D                    # K is zero-padded to lay.in_ld (a multiple of the 64-wide K tile)
        us = time_graph(lambda: eng.gemm(x, lay.in_ld, lay.WxT.data_ptr(), lay.in_ld, lw['Gx'].data_ptr(), lay.N4, M, lay.N4,
                                         lay.in_ld, bias=lay.bias_ptr(eng.store.p)), 20)
        flops = 2 * M * N * K
        ach = flops / (us * 1e-6) / 1e12
        alg_bytes = 2 * M * K + 2 * N * K + 4 * M * N
        # HBM-side bytes per launch of THIS instance from the committed PMC run (profiles/r01f_pmc_gemm_gx.json:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/roofline_gemm.py, FETCH_SIZE x2 as
        # MI355X_MICROARCH.md prescribes); a measured constant of this round, not live
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01f_pmc_gemm_gx.json')) as f:
                t = json.load(f)['k_gemm_nt']
            if args.config == 'cfg2' and B == 256:
                traffic = t['hbm_read_bytes'] + t['hbm_write_bytes']
        except Exception:
            pass
        roof = dict(bound='mfma', kernel='k_gemm_nt', instance='encoder input projection M=%d N=%d K=%d (bias epilogue, fp32 out)' % (M, N, K),
                    achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s', frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    traffic=traffic, us_per_launch=round(us, 2), flops_per_launch=flops, algorithmic_hbm_bytes_per_launch=alg_bytes)
        # the recurrences (second largest share): one persistent launch per layer and direction pair, time per step
        d = lay.desc(lw, True)
        if eng.persistent_fwd and lay.persistent_ok(B, eng.num_cus):
            us_f = time_graph(lambda: lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, S)), 3) / S
            extra['lstm_fwd_us_per_step'] = round(us_f, 3)
        if eng.persistent_bwd and lay.persistent_bwd_ok(B, eng.num_cus):
            us_b = time_graph(lambda: lay.bwd_rec(lw, x, ws['lens_d'], ws['dY'][1].data_ptr(), lay.ldy, True, None, 0,
                                                  dy_masked=lay.out_drop(True) is not None), 3) / S
            extra['lstm_bwd_us_per_step'] = round(us_b, 3)
        extra['lstm_step_flops'] = 2 * B * lay.H * 4 * lay.H * 2      # both directions, one time step of one layer
        assert int(eng.sync_err[0].item()) == 0

    if rank == 0:
        utt = B * world * args.steps / el
        S = ceil_div(T, spec.decimation)
        rec = 3 * recurrent_flops_fwd(spec_kw, S, L) * utt / 1e12
        out = dict(metric='train utterances/sec', value=round(utt, 2), unit='utterances/s', n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=round(1e3 * el / args.steps, 4), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype='bf16', data='synthetic',
                   config=dict(workload='%s: 1 subject, %d electrodes x %d samples, B=%d/GPU, conv%d -> %dx biLSTM(%d) -> LSTM(%d) -> %d words, L=%d, Adam+EMA'
                               % (args.config, spec_kw['channels'][sid], T, B, spec.enc_embed, len(spec.enc_rnn), spec.enc_rnn[0],
                                  spec.dec_rnn, spec.vocab, L), global_batch=B * world, parallelism='dp%d' % world,
                               hipgraph=not args.no_graph),
                   recurrent_gemm_tflops=round(rec, 3), recurrent_gemm_frac_of_peak=round(rec / MFMA_BF16_PEAK_TFLOPS / world, 5),
                   final_loss=round(losses['total'], 4), recurrence=extra, roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(spec_kw, T, L)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
