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
every rank steps its own B utterances, gradients are all-reduced every step (the collectives
are nodes of the step's one hipGraph).  Prints ONE JSON line on rank 0:
  value / ms_per_step   whole-job utterances/s and step time over the timed region;
  roofline              the dominant rocprof row (K-major grouped GEMM): `frac` / `achieved` IN-STEP, measured live through
                        the kernels' stamp hook over ten replays of a stamped step graph; `frac_isolated` = the same launches
                        replayed alone between HIP events; `frac_in_step_profile` = priced at the committed rocprofv3 row;
                        `traffic` = HBM bytes per launch of the committed PMC pass; roofline_all_gemm_instances: every instance;
  recurrence / recurrent_gemm_frac_of_peak   per-step time of the persistent recurrences, the north-star fraction;
  decode                greedy (eager / graph-replayed) and beam-4 decode of the batch, ms per batch;
  configs               cfg3 / cfg4 / cfg5 (fp32 and bf16-staged inputs) measured in the same process (N = 1 only);
  cpu_baseline          the same train step on the box's host cores (N = 1 only): the C++17 / OpenMP fp32 restatement
                        (oracle/cpu_step.cpp); `torch_port` = the torch-CPU model beside it (d5 (ii)).
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
    # cfg3 (BASELINE.json configs[2]): 4 participants with their own conv front-ends (grids 16x16, 16x16, 8x16, 16x16:
    # mochastar_word_sequence.yaml:57-59, 150-152, 243-245, 336-338), one participant per step in turn
    'cfg3': (dict(channels={400: 256, 401: 256, 402: 128, 403: 256}, decimation=12, enc_embed=100, enc_rnn=[400, 400, 400],
                  dec_embed=150, dec_rnn=800, vocab=1806, aux_layer=1, aux_hidden=[225], aux_dim=13, ff_dropout=0.1,
                  rnn_dropout=0.5), 256, 400, 10),
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


def cpu_baseline_cxx(cfg, batch_override, timed=5):
    """oracle/cpu_step.cpp -- this repository's C++17 / OpenMP fp32 implementation of the identical train step
    (pinned against the NumPy oracle by tests/test_cpu_step.py) -- built with g++ -march=native ON this box and timed in a process
    of its own (its OpenMP pool does not share the cores with torch's): warm-up, one step at each of nproc/4, nproc/2, nproc
    threads, then `timed` steps at the best count, median.  None if it cannot be built or run here."""
    import subprocess
    if batch_override:
        return None
    try:
        r = subprocess.run([sys.executable, '-m', 'oracle.cpu_step', cfg, str(timed)], cwd=ROOT, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        return json.loads(line)
    except Exception as e:                     # no g++, a build failure, a time-out: the torch port alone is reported
        sys.stderr.write('cpu_baseline: the C++ step did not run (%r)\n' % (e,))
        return None


def cpu_baseline(spec_kw, B, T, L, timed=5, cfg=None, batch_override=None):
    cxx = cpu_baseline_cxx(cfg, batch_override) if cfg else None
    port = cpu_baseline_torch(spec_kw, B, T, L, timed=2 if cxx else timed, counts=(16,) if cxx else (8, 16, 32, 64))
    if not cxx:
        return port
    return dict(value=cxx['value'], unit='utterances/s', cores=int(cxx['cores']), kind='port',
                sample='C++17/OpenMP fp32 restatement of the train step (oracle/cpu_step.cpp: packed-panel SGEMM, per-step recurrent GEMMs, '
                       'manual reverse mode, Adam+EMA, dropout on), g++ -O3 -march=native on this box, B=%d T=%d; thread sweep (threads: s/step): %s; '
                       'then %d timed steps at %d threads, median %.3f s/step (min %.3f, max %.3f); usable hardware threads (affinity mask capped by the cgroup quota) %d of %d listed.  torch_port: %s'
                       % (cxx['B'], cxx['T'], ', '.join('%d: %.3f' % (n, t) for n, t in cxx['sweep']), timed, cxx['cores'], cxx['s_per_step'],
                          cxx['min'], cxx['max'], cxx['nproc'], cxx.get('listed', cxx['nproc']), port['sample']),
                torch_port=dict(value=port['value'], cores=port['cores']))


def cpu_baseline_torch(spec_kw, B, T, L, timed=5, counts=(8, 16, 32, 64)):
    """the CPU number timed beside the GPU run.  The reference's own CPU path (TF1.x + un-vendored
    packages) cannot run here, so this is `oracle/torch_model.py` -- the independent torch-CPU implementation of the
    same architecture (torch.nn.LSTM oneDNN/MKL kernels, fp32, autograd, Adam + EMA, dropout on) -- on the SAME batch
    size and shapes as the GPU step (kind "port").  Bounded sample (about 30 s of CPU work): one warm-up step, ONE timed
    step at each of 8 / 16 / 32 / 64 threads (all of them: torch's default of every hardware thread is an order of
    magnitude slower for a recurrence of small per-step GEMMs, and which count is best differs from box to box), then
    `timed` steps at the best count, median; `cores` = that count, the sweep is reported in `sample`."""
    import torch
    from oracle import seq2seq as O
    from oracle.torch_model import train_step_fn
    sid = list(spec_kw['channels'])[0]
    kw1 = dict(spec_kw, channels={sid: spec_kw['channels'][sid]})
    step, _ = train_step_fn(O.NetSpec(**kw1), synth_batch(kw1, B, T, L, seed=1))
    ncpu = os.cpu_count() or 1
    counts = [n for n in counts if n <= ncpu] or [ncpu]
    torch.set_num_threads(counts[min(1, len(counts) - 1)])
    step()                                        # warm-up (allocations, oneDNN primitive caches)
    sweep = []
    for nt in counts:
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step()
        sweep.append((time.perf_counter() - t0, nt))
    threads = min(sweep)[1]
    torch.set_num_threads(threads)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    cpu = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), '')
    except OSError:
        pass
    return dict(value=round(B / med, 3), unit='utterances/s', cores=int(threads), kind='port',
                sample='torch-CPU fp32 model of the same architecture (oracle/torch_model.py: nn.LSTM, conv1d, autograd, Adam+EMA), '
                       'B=%d T=%d; thread sweep (s/step): %s; then %d timed train steps at %d threads, median %.3f s/step '
                       '(min %.3f, max %.3f); host: %s, nproc=%d'
                       % (B, T, ', '.join('%d: %.2f' % (n, t) for t, n in sweep), len(ts), threads, med, min(ts), max(ts), cpu, ncpu))


# in-step averages (rocprofv3 --kernel-trace --stats of the train step) and HBM bytes (PMC passes) of the GEMM instances, from the
# committed profile of the round: profiles/roofline_refs.json (scripts/make_roofline_refs.py writes it, with its sources)
def roofline_refs():
    try:
        with open(os.path.join(ROOT, 'profiles', 'roofline_refs.json')) as f:
            return json.load(f)
    except Exception:
        return {}


GEMM_KERNELS = {'tn128g': 'k_gemm_tn_group (K-major operands: the weight gradients of a backward stage in one grouped launch, 128x128 tiles)',
                'tn256': 'k_gemm_nt<256,256,2,4,false,true> (K-major operands, both output dimensions >= 1024)',
                'tn128': 'k_gemm_nt<128,128,2,2,false,true> (K-major operands: weight gradients and the ragged edge of the large ones, lean epilogue)',
                'nt128': 'k_gemm_nt<128,128,2,2,true,false> (K-contiguous, full epilogue)',
                'nt256': 'k_gemm_nt<256,256,2,4,false,false> (K-contiguous, large plain products)'}


def synth_batch_device(spec_kw, sid, B, T, L, seed, device):
    """The same synthetic workload as synth_batch(), generated ON the device (the other BASELINE configs of the default
    line: cfg5's batch is 2.1 GB of fp32, which numpy takes tens of seconds to draw): |N(0,1)| standardised + a
    sentence-dependent low-rank signal, 50 sentences of 3..L-1 words + <EOS>, N(0,1) auxiliary targets."""
    import torch
    g = torch.Generator(device=device).manual_seed(int(seed))
    C = spec_kw['channels'][sid]
    V, K = spec_kw['vocab'], spec_kw['aux_dim']
    rng = np.random.default_rng(seed)
    nsent = 50
    sents = [rng.integers(3, V, size=n) for n in rng.integers(3, L, size=nsent)]
    which = rng.integers(0, nsent, size=B)
    X = torch.randn(B, T, C, generator=g, device=device).abs_()
    X.sub_(X.mean()).div_(X.std())
    basis = torch.randn(nsent, 2, C, generator=g, device=device)
    tt = torch.linspace(0, 1, T, device=device)[:, None]
    wd = torch.as_tensor(which, device=device)
    freq = (1 + wd % 5).to(torch.float32)[:, None, None]
    X.add_(0.5 * (torch.sin(2 * np.pi * freq * tt[None]) * basis[wd, 0][:, None, :] + tt[None] * basis[wd, 1][:, None, :]))
    X[X == 0] = 1e-3
    Y = np.zeros((B, L), np.int32)
    for b in range(B):
        w = sents[which[b]]
        Y[b, :len(w)] = w
        Y[b, len(w)] = 1
    A = torch.randn(B, T, K, generator=g, device=device)
    return X, torch.as_tensor(Y, device=device), A


def measure(cfg_name, args, steps, warmup, rank, world, dev_index, sync_factory, roofline, device_data=False, batch=None, inputs='fp32'):
    """One configuration: engine, synthetic batch resident in HBM, `warmup` untimed + `steps` timed train steps (barrier +
    synchronize on both sides, max over ranks), and -- on rank 0 -- the live roofline legs.  Returns the dict of the line."""
    import torch
    from ecog2txt_amd.engine import Seq2SeqEngine, NetSpec, ceil_div, capture
    spec_kw, B, T, L = CONFIGS[cfg_name]
    B = batch or B
    spec = NetSpec(**spec_kw)
    device = 'cuda:%d' % dev_index
    eng = Seq2SeqEngine(spec, device=device, seed=1234 + rank, options=dict(kv.split('=', 1) for kv in args.engine_option))
    eng.init_params(seed=0)
    sync = sync_factory(eng) if sync_factory is not None else None
    if sync is not None:
        sync.broadcast_([eng.store.p, eng.store.ema])
    eng.pack('p')
    # one workspace per participant (cfg3: four, stepped in turn -- SURVEY.md 8 d2); synthetic batches resident in HBM
    sids = list(spec.channels)
    wss = []
    for i, sid in enumerate(sids):
        ws = eng.workspace(sid, B, T, L)
        if device_data:
            X, Y, A = synth_batch_device(spec_kw, sid, B, T, L, 100 + rank + 17 * i, device)
            ws['X'].copy_(X); ws['Y'].copy_(Y); ws['auxT'].copy_(A)
            del X, A
            cnt_src = (Y.cpu().numpy(), None)
        else:
            batch_ = synth_batch(dict(spec_kw, channels={sid: spec.channels[sid]}), B, T, L, seed=100 + rank + 17 * i)
            eng.set_batch(ws, batch_)
            cnt_src = (batch_['decoder_targets'], batch_['encoder_targets'])
        if sync is not None:
            # losses are normalised by the GLOBAL token counts, so the exchange is a plain sum (parallel.py)
            cnt = sync.allreduce_numpy(np.array(eng.local_counts(*cnt_src), np.int64))
            eng.set_global_counts(ws, int(cnt[0]), int(cnt[1]))
        wss.append(ws)
    torch.cuda.synchronize()
    staging_ms = None
    if inputs == 'bf16':
        # SURVEY.md 8 d4 "bf16 in": the inputs are staged ONCE (per fit) as the bf16 im2row rows of the front-end
        # (Seq2SeqEngine.pack_inputs, timed here) and stay resident in that form; a step's operand is assembled from them by a
        # blocked row gather (load_packed_batch; outside the timed region, like the gather of the fp32 form)
        t0 = time.perf_counter()
        for ws in wss:
            ws['_pk'] = eng.pack_inputs(ws['sid'], ws['X'])
        torch.cuda.synchronize()
        staging_ms = 1e3 * (time.perf_counter() - t0)
        ident = torch.arange(B, dtype=torch.int32, device=device)
        for ws in wss:
            eng.load_packed_batch(ws, ws.pop('_pk'), ident)
            ws['X'].fill_(float('nan'))                     # (no step may read x from here on)
        torch.cuda.synchronize()
    it = [0]

    def step():
        eng.train_step(wss[it[0] % len(wss)], use_graph=not args.no_graph, sync=sync)
        it[0] += 1

    # the step loop runs with the engine's own stream current, as SequenceNetwork.fit runs it (engine.on_step_stream)
    with eng.on_step_stream():
        for _ in range(max(warmup, len(wss))):
            step()
        torch.cuda.synchronize()
        if sync is not None:
            sync.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    if sync is not None:
        sync.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if sync is not None:
        els = np.zeros(world, np.float32)
        els[rank] = el
        el = float(sync.allreduce_numpy(els).max())           # max over ranks
    ws = wss[0]
    losses = eng.losses(wss[(it[0] - 1) % len(wss)])
    assert np.isfinite(losses['total']), losses

    # ---- per-kernel rooflines, measured live.  The dominant kernel of the step (rocprofv3 summary under profiles/) is the
    #      MFMA GEMM, in several instances.  Every product of one eager step is logged with the instance the library picks
    #      for it and its ALGORITHMIC flops (2*M*N*K on the unpadded dimensions); each instance's
    #      launches are then replayed back to back from a hipGraph on the bench stream, bracketed by HIP events on THAT
    #      stream: achieved = sum of flops / sum of launch durations = flops per launch / average launch duration.
    #      (In the step the same launches share the chip with the side stream; the in-step averages are in profiles/.)
    roof, groups, extra = None, {}, {}
    S = ceil_div(T, spec.decimation)
    if rank == 0 and roofline:
        def time_graph(fn, reps):
            fn(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with capture(gr):
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

        eng._gemm_log = []
        eng.forward(ws, train=True)
        eng.backward(ws, train=True)
        torch.cuda.synchronize()
        log, eng._gemm_log = eng._gemm_log, None
        refs = roofline_refs().get(cfg_name if B == CONFIGS[cfg_name][1] else '', {})
        for inst in ('tn128g', 'tn128', 'tn256', 'nt128', 'nt256'):
            recs = [r for r in log if r['inst'] == inst]
            if not recs:
                continue
            us = time_graph(lambda: [eng.gemm_replay(r) for r in recs], 3)       # one pass over all launches of the instance
            flops = sum(r['flops'] for r in recs)
            big = max(recs, key=lambda r: r['flops'])
            tf = flops / (us * 1e-6) / 1e12
            ref = refs.get(inst, {})
            in_step_us = ref.get('in_step_us_per_launch')
            groups[inst] = dict(bound='mfma', kernel=GEMM_KERNELS[inst], launches_per_step=len(recs),
                                achieved=round(tf, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                                # the same launches INSIDE the step share the chip with the other stream: fraction from the committed
                                # rocprofv3 row of the round (in-step average duration of this kernel), beside the isolated `frac`
                                frac_in_step_profile=(round(flops / len(recs) / (in_step_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if in_step_us else None),
                                in_step_profile_us_per_launch=in_step_us, in_step_source=ref.get('in_step_source'),
                                traffic=ref.get('hbm_bytes_per_launch'), traffic_source=ref.get('traffic_source'),
                                us_per_launch=round(us / len(recs), 2), flops_per_launch=int(flops / len(recs)),
                                algorithmic_hbm_bytes_per_launch=int(sum(r['in_bytes'] + r['out_bytes'] for r in recs) / len(recs)),
                                us_per_step=round(us, 1), includes_splitk_reduce=any(r['splits'] != 1 for r in recs),
                                largest=(big.get('desc') or 'M=%d N=%d K=%d x%d (splits %d)' % (big['M'], big['N'], big['K'], big['batch'], big['splits'])))
        # ---- the same launches INSIDE the step, measured live: with the stamp hook on (e2t_gemm_stamps) every GEMM launch of a freshly
        #      captured step records [first workgroup start, last workgroup end] on the chip-wide 100-MHz clock; ten replays, the
        #      stamps read and reset after each: in-step average duration per instance = what `rocprofv3 --kernel-trace --stats` of
        #      this command reports for the kernel row (profiles/: the committed summary of the same command)
        live = {}
        try:
            NSL = 512
            stamps = torch.empty(2 * NSL, dtype=torch.int64, device=device)
            blank = torch.tensor([-1, 0] * NSL, dtype=torch.int64, device=device)       # ~0 / 0 as unsigned words
            saved = [w['graph'] for w in wss]
            for w in wss:
                w['graph'] = {}
            lib_ = __import__('ecog2txt_amd.hip_lib', fromlist=['lib']).lib
            lib_.e2t_gemm_stamps(stamps.data_ptr(), NSL)
            import ctypes as C_
            names = {0: 'tn128g', 1: 'tn128', 2: 'tn256', 3: 'nt128', 4: 'nt256'}
            acc = {}
            with eng.on_step_stream():
                for _ in range(2 * len(wss)):
                    step()                                  # eager warm-up + capture with stamp slots, one replay
                torch.cuda.synchronize()
                for _ in range(10):
                    stamps.copy_(blank)
                    eng.train_step(wss[0], use_graph=not args.no_graph, sync=sync)
                    torch.cuda.synchronize()
                    kinds = (C_.c_int * NSL)()
                    lib_.e2t_gemm_stamp_kinds(kinds, NSL)
                    st_ = stamps.cpu().numpy().view(np.uint64).reshape(NSL, 2)
                    for k in range(NSL):
                        if kinds[k] >= 0 and st_[k, 1] > 0 and st_[k, 0] != np.uint64(0xFFFFFFFFFFFFFFFF):
                            acc.setdefault(names[kinds[k]], []).append((int(st_[k, 1]) - int(st_[k, 0])) * 0.01)     # 100 MHz -> us
            lib_.e2t_gemm_stamps(None, 0)
            for w, g_ in zip(wss, saved):
                w['graph'] = {}                              # (graphs captured with stamp slots must not outlive the buffer)
            live = {k: (float(np.mean(v)), len(v) // 10) for k, v in acc.items()}
        except Exception as e:
            print('bench: live in-step timing unavailable (%r)' % (e,), file=sys.stderr)
            try:
                __import__('ecog2txt_amd.hip_lib', fromlist=['lib']).lib.e2t_gemm_stamps(None, 0)
            except Exception:
                pass
        for inst, (us_l, n_l) in live.items():
            if inst in groups:
                gq = groups[inst]
                # a product the library cuts into two launches (ragged edge, last round) counts as two stamped launches of one
                # logged product: the in-step time of the instance per step is what matters
                us_step = us_l * n_l
                gq['in_step_live_us_per_step'] = round(us_step, 1)
                gq['in_step_live_us_per_launch'] = round(us_step / gq['launches_per_step'], 2)
                tf_l = gq['flops_per_launch'] * gq['launches_per_step'] / (us_step * 1e-6) / 1e12
                # `achieved` / `frac` = IN-STEP, measured live (they agree with the kernel row of the committed rocprofv3 summary of
                # this command: `frac_in_step_profile`); the isolated replay keeps its own names
                gq.update(achieved_isolated=gq['achieved'], frac_isolated=gq['frac'], achieved=round(tf_l, 2),
                          frac=round(tf_l / MFMA_BF16_PEAK_TFLOPS, 4), measured='in-step, live (e2t_gemm_stamps: first workgroup start .. last workgroup end of every launch of 10 replayed steps)')
        if args.gemm_detail:
            for r in log:
                us1 = time_graph(lambda: eng.gemm_replay(r), 10)
                print('%-6s M=%5d N=%5d K=%5d x%d splits %d %s  %7.1f us  %6.1f TF  %s' % (r['inst'], r['M'], r['N'], r['K'], r['batch'], r['splits'],
                      'side' if r['side'] else 'main', us1, r['flops'] / us1 / 1e6, r.get('desc', '')), file=sys.stderr)
        if groups:
            dom = max(groups, key=lambda k: groups[k]['us_per_step'])          # the instance with the largest share of the step
            roof = dict(groups[dom], instance=dom)
        # the recurrences (second largest share): one persistent launch per layer and direction pair, time per step
        lay, lw = eng.enc[1 if len(eng.enc) > 1 else 0], ws['enc'][1 if len(eng.enc) > 1 else 0]
        li = 1 if len(eng.enc) > 1 else 0
        x = ws['enc'][li - 1]['Ydrop'].data_ptr() if li > 0 else ws['E'].data_ptr()
        fwd_persist = eng.persistent_fwd and lay.persistent_ok(B, eng.num_cus)
        bwd_persist = eng.persistent_bwd and lay.persistent_bwd_ok(B, eng.num_cus)
        us_f = time_graph(lambda: lay.fwd(lw, x, ws['lens_d'], eng.store.p, True, steps=(0, S)), 3) / S
        us_b = time_graph(lambda: lay.bwd_rec(lw, x, ws['lens_d'], ws['dY'][li].data_ptr(), lay.ldy, True, None, 0,
                                              dy_masked=lay.out_drop(True) is not None), 3) / S
        fl = 2 * B * lay.H * 4 * lay.H * 2           # both directions, one time step of one layer
        extra = dict(lstm_fwd_us_per_step=round(us_f, 3), lstm_bwd_us_per_step=round(us_b, 3), lstm_step_flops=fl,
                     lstm_fwd_persistent=bool(fwd_persist), lstm_bwd_persistent=bool(bwd_persist),
                     lstm_fwd_frac_of_mfma_peak=round(fl / (us_f * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     lstm_bwd_frac_of_mfma_peak=round(fl / (us_b * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4))
        eng.check_sync()
        # ---- decoding (BASELINE.json's metric also names decode WER; the sequences themselves are checked against the oracle in
        #      tests/): greedy decode of the same batch with the EMA weights -- encoder + L decoder steps --, eager and as one
        #      captured graph, and beam search of width 4 (eager, B x 4 hypothesis rows)
        if inputs == 'fp32':
            def timed(fn, reps=5):
                fn(); torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t0_) / reps
            try:
                ms_e = timed(lambda: eng.greedy_decode(ws, which='ema'))
                ms_g = timed(lambda: eng.greedy_decode(ws, which='ema', use_graph=True))
                ms_b = timed(lambda: eng.beam_decode(ws, 4, which='ema'), reps=3)
                extra['decode'] = dict(greedy_ms_per_batch_eager=round(ms_e, 3), greedy_ms_per_batch_graph=round(ms_g, 3),
                                       greedy_utterances_per_s=round(B / (min(ms_e, ms_g) * 1e-3), 1), beam4_ms_per_batch=round(ms_b, 3),
                                       max_tokens=L, weights='ema')
            except Exception as e:                             # (never let the decode leg take the training line down)
                extra['decode'] = dict(error=repr(e)[:200])
            eng.pack('p')

    out = None
    if rank == 0:
        utt = B * world * steps / el
        rec = 3 * recurrent_flops_fwd(spec_kw, S, L) * utt / 1e12
        chans = '/'.join(str(c) for c in spec.channels.values())
        out = dict(metric='train utterances/sec', value=round(utt, 2), unit='utterances/s', n_gpus=world, steps=steps,
                   warmup=warmup, ms_per_step=round(1e3 * el / steps, 4), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype='bf16', data='synthetic',
                   config=dict(workload='%s: %d subject(s), %s electrodes x %d samples (%s), B=%d/GPU, conv%d -> %dx biLSTM(%d) -> LSTM(%d) -> %d words, L=%d, Adam+EMA'
                               % (cfg_name, len(sids), chans, T,
                                  'fp32 inputs resident in HBM' if inputs == 'fp32' else
                                  'inputs resident in HBM as bf16 im2row rows, staged once from fp32 in %.1f ms' % staging_ms,
                                  B, spec.enc_embed, len(spec.enc_rnn), spec.enc_rnn[0],
                                  spec.dec_rnn, spec.vocab, L), inputs=inputs, global_batch=B * world, parallelism='dp%d' % world,
                               hipgraph=not args.no_graph, exchange=(type(sync).__name__ if sync is not None else None),
                               # how many ranks the communicator itself saw (e2t_comm_size; torch.distributed's world for the other transport)
                               communicator_ranks=(int(sync.lib.e2t_comm_size(sync.comm)) if hasattr(sync, 'comm') else getattr(sync, 'world', 1))),
                   recurrent_gemm_tflops=round(rec, 3), recurrent_gemm_frac_of_peak=round(rec / MFMA_BF16_PEAK_TFLOPS / world, 5),
                   final_loss=round(losses['total'], 4), decode=extra.pop('decode', None), recurrence=extra, roofline=roof,
                   roofline_all_gemm_instances=groups)
    # release the engine's device memory before the next configuration (cfg5's workspace is ~9 GB)
    eng._ws.clear()
    del eng, wss, ws
    import gc as _gc
    _gc.collect()
    torch.cuda.empty_cache()
    return out, sync


def _opt_true(opts, name):
    return any(kv.split('=', 1)[0] == name and kv.split('=', 1)[1].lower() in ('1', 'true') for kv in opts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='cfg2', choices=list(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU (default: the config\'s 256)')
    ap.add_argument('--inputs', default='fp32', choices=['fp32', 'bf16'],
                    help='how the inputs are resident in HBM: fp32 [B][T][C] (default, the headline form) or the bf16 im2row rows of the front-end (SURVEY 8 d4)')
    ap.add_argument('--engine-option', action='append', default=[], metavar='KEY=VALUE',
                    help='override of Seq2SeqEngine.OPTIONS (diagnostics), e.g. persistent=0; may be repeated')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the `configs` block (cfg3 / cfg4 / cfg5 measured in the same process)')
    ap.add_argument('--gemm-detail', action='store_true', help='stderr: isolated time of every logged product of the step')
    args = ap.parse_args()

    import torch
    from ecog2txt_amd import parallel

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU) -- and fail rather than report
        # one GPU N times when the box has fewer (E2T_BENCH_BACKEND=gloo: the control-flow diagnostic with several ranks per GPU)
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get('E2T_BENCH_BACKEND') != 'gloo':
            sys.exit('bench.py: --gpus %d, but %d GPU(s) are visible' % (args.gpus, ndev))
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d, but the launcher started %d rank(s) (torch.distributed.run --nproc-per-node %d)' % (args.gpus, world, args.gpus))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)        # (diagnostics: several ranks on one GPU with E2T_BENCH_BACKEND=gloo)
    torch.cuda.set_device(dev_index)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    # data parallel: RCCL through the C ABI (e2t_comm_*); E2T_COMM=torch selects torch.distributed's "nccl" instead
    def sync_factory(eng):
        if os.environ.get('E2T_COMM', 'rccl') == 'rccl':
            try:
                return parallel.make_sync(eng.store.g)
            except Exception as e:                          # (symmetric on all ranks of a node: same library, same driver)
                print('bench: direct RCCL exchange unavailable (%r); falling back to torch.distributed "nccl"' % (e,), file=sys.stderr)
                os.environ['E2T_COMM'] = 'torch'
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get('E2T_BENCH_BACKEND', 'nccl')
            dist.init_process_group(backend, **({'device_id': torch.device('cuda', dev_index)} if backend == 'nccl' else {}))
        return parallel.make_sync(eng.store.g)

    done = printed = None
    if world > 1 and ndev < world and os.environ.get('E2T_BENCH_BACKEND') != 'gloo':
        sys.exit('bench.py: %d ranks, but %d GPU(s) are visible' % (world, ndev))
    out, sync = measure(args.config, args, args.steps, args.warmup, rank, world, dev_index, sync_factory if world > 1 else None,
                        roofline=not args.no_roofline, batch=args.batch, inputs=args.inputs)
    if world > 1 and rank == 0:
        out['config']['schedule'] = 'dp_one_graph' if _opt_true(args.engine_option, 'dp_one_graph') else 'graph per backward stage, eager collectives'
    if world > 1 and not args.no_graph and not any(kv.startswith('dp_one_graph=') for kv in args.engine_option) \
            and getattr(sync, 'capturable', False) and os.environ.get('E2T_BENCH_ONE_GRAPH', '0') == '1':
        # The data-parallel step as ONE graph with the RCCL collectives as nodes (engine option dp_one_graph; 10 % faster than the
        # graph-per-stage schedule on a one-rank communicator) has never run with real peers on the builder's side.  The line above
        # is safe; this second measurement is OPT-IN (E2T_BENCH_ONE_GRAPH=1) and runs under a watchdog: should a replay of captured
        # collectives hang, rank 0 prints the line it has and every rank leaves WITH A NON-ZERO STATUS.  `value` is always the
        # default schedule's; the other one is reported beside it under config.schedules_measured.
        import threading
        first = out
        done, printed = threading.Event(), [False]
        limit = float(os.environ.get('E2T_BENCH_WATCHDOG_S', '300'))

        def watchdog():
            if not done.wait(limit if rank == 0 else limit + 20.0):          # (rank 0 prints before the others leave)
                if rank == 0 and not printed[0]:
                    first['config']['dp_one_graph'] = 'no result within %.0f s (watchdog); the line is the graph-per-stage schedule' % limit
                    print(json.dumps(first), flush=True)
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        sync.barrier()
        sync.close()
        args.engine_option = list(args.engine_option) + ['dp_one_graph=True']
        out2, err2 = None, None
        try:
            out2, sync = measure(args.config, args, args.steps, args.warmup, rank, world, dev_index, sync_factory, roofline=False,
                                 batch=args.batch, inputs=args.inputs)
        except Exception as e:                               # (the line of the default schedule must survive whatever happens here;
            err2, sync = repr(e)[:300], None                 #  ranks left waiting in a collective are released by their watchdogs)
        # (the watchdog stays armed until the closing barrier at the end of main(): a rank that failed here leaves, and the others
        #  must not wait for it for ever)
        if err2 is not None:
            if rank == 0:
                first['config']['dp_one_graph'] = 'failed: %s; the line is the graph-per-stage schedule' % err2
                print(json.dumps(first), flush=True)
            os._exit(3)
        if rank == 0:
            first['config']['schedules_measured'] = dict(graph_per_stage_ms=first['ms_per_step'], one_graph_ms=out2['ms_per_step'],
                                                         one_graph_value=out2['value'], one_graph_final_loss=out2['final_loss'])
            out = first
    if rank == 0:
        # The other BASELINE.json configurations, measured in the SAME process on one GPU (10 warm-up + 20 timed steps each,
        # synthetic inputs generated on the device): cfg3 (4 participants in turn), cfg4 (wide model), cfg5 (long / wide input).
        if world == 1 and args.config == 'cfg2' and args.batch is None and not args.no_configs:
            block = {}
            for c in ('cfg3', 'cfg4', 'cfg5', 'cfg5_bf16_staged'):
                t0 = time.perf_counter()
                try:
                    o, _ = measure(c.replace('_bf16_staged', ''), args, 20, 10, 0, 1, dev_index, None, roofline=not args.no_roofline, device_data=True,
                                   inputs='bf16' if c.endswith('_bf16_staged') else 'fp32')
                except Exception as e:                     # the headline line must not die with a side configuration
                    block[c] = dict(error=repr(e)[:300])
                    continue
                dom = o['roofline'] or {}
                block[c] = dict(workload=o['config']['workload'], ms_per_step=o['ms_per_step'], value=o['value'], unit=o['unit'],
                                steps=20, warmup=10, recurrent_gemm_frac_of_peak=o['recurrent_gemm_frac_of_peak'],
                                dominant=dict(instance=dom.get('instance'), kernel=dom.get('kernel'), frac=dom.get('frac'), measured=dom.get('measured'),
                                              frac_isolated=dom.get('frac_isolated'), frac_in_step_profile=dom.get('frac_in_step_profile'), in_step_source=dom.get('in_step_source'),
                                              in_step_live_us_per_launch=dom.get('in_step_live_us_per_launch'),
                                              us_per_launch=dom.get('us_per_launch'), flops_per_launch=dom.get('flops_per_launch'),
                                              largest=dom.get('largest')),
                                gemm_instances={k: dict(frac=v['frac'], frac_isolated=v.get('frac_isolated'), us_per_step_isolated=v['us_per_step'],
                                                        us_per_step_in_step=v.get('in_step_live_us_per_step'))
                                                for k, v in o['roofline_all_gemm_instances'].items()},
                                recurrence=o['recurrence'], final_loss=o['final_loss'], wall_s=round(time.perf_counter() - t0, 1))
            out['configs'] = block
        if world == 1 and not args.no_cpu_baseline:
            spec_kw, B, T, L = CONFIGS[args.config]
            out['cpu_baseline'] = cpu_baseline(spec_kw, args.batch or B, T, L, cfg=args.config, batch_override=args.batch)
        print(json.dumps(out), flush=True)
        if printed is not None:
            printed[0] = True
    if sync is not None and hasattr(sync, 'close'):
        sync.barrier()
        sync.close()
    if done is not None:
        done.set()


if __name__ == '__main__':
    main()
