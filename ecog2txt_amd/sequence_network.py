"""`SequenceNetwork`: the MI355X backend object that `MultiSubjectTrainer.net` holds.

Drop-in for `machine_learning.neural_networks.sequence_networks.SequenceNetwork` as the
reference uses it (construction ecog2txt/trainers.py:126-135; `fit` :318/:355/:367 with the
kwargs of :308-314 and :341-366; `restore_and_assess` :379-380; `restore_and_get_saliencies`
:722-725; `get_weights_as_numpy_array` :699-700, :750-751; attributes read or written by the
trainer: checkpoint_path :219, N_epochs :324/:351/:365, layer_sizes :385/:576,
TEMPORALLY_CONVOLVE :386, EMA_decay :387, FF_dropout/RNN_dropout :572-573,
assessment_epoch_interval :583, assessment_GPU :856, TARGETS_ARE_SEQUENCES :292).
All device work goes through ecog2txt_amd.engine (hand-written HIP kernels); host code here is
data staging, the epoch loop, metrics and checkpoint files."""
import os
import re
import time

import numpy as np

from . import tfrecord
from .toolbox import auto_attribute, wer_vector, MutableNamedTuple

EMA_SUFFIX = '/ExponentialMovingAverage'        # trainers.py:467-468


class AssessmentTuple(MutableNamedTuple):
    """Per-partition results; the fields the trainer/plotters read (trainers.py:584-596, 610;
    plotters.py:636)."""
    __slots__ = ['decoder_accuracies', 'decoder_word_error_rates', 'decoder_confusions', 'accuracy',
                 'word_error_rate', 'hypotheses', 'references', 'losses']


def target_inds_to_sequences(hypotheses, targets_list):
    """index rows -> text: join tokens, '_' -> ' ', drop <pad>/<EOS>, rstrip (trainers.py:952-963)."""
    from . import pad_token, EOS_token
    return [''.join(targets_list[i] for i in row).replace('_', ' ').replace(pad_token, '').replace(EOS_token, '').rstrip()
            for row in hypotheses]


def load_examples(subject, blocks):
    """Parse the TFRecords of `blocks` into network-ready arrays using the subject's data manifests
    (role of tfh.parse_protobuf_seq2seq_example, reference trainers.py:896-900, subjects.py:216-218)."""
    dms = subject.data_manifests
    out = []
    for blk in sorted(blocks):
        path = subject.tf_record_partial_path.format(blk)
        for payload in tfrecord.tf_record_iterator(path):
            raw = tfrecord.decode_example(payload)
            ex = {}
            for key, dm in dms.items():
                v = raw[dm.sequence_type]
                if dm.is_continuous:
                    v = np.asarray(v, np.float32).reshape(-1, dm.num_features_raw)     # stored flattened (trainers.py:865)
                ex[key] = dm.transform(v)
            out.append(ex)
    return out


class _Arrays(dict):
    """dict of arrays with the `.files` attribute of an NpzFile"""
    @property
    def files(self):
        return list(self)


def load_checkpoint_arrays(prefix, names=None):
    """Variables of checkpoint `prefix`: this backend's `.npz` (weights, EMA shadows, Adam state) when present, else
    a TensorFlow V2 checkpoint `prefix.index` + `prefix.data-*` as written by the reference's TF1 Saver."""
    if os.path.exists(prefix + '.npz'):
        return np.load(prefix + '.npz')
    from . import tf_checkpoint
    return _Arrays(tf_checkpoint.read_checkpoint(prefix, names=names))


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except OSError:                      # (EPERM: it exists and belongs to someone else)
        return True
    return True


class SequenceNetwork:
    @auto_attribute(CHECK_MANIFEST=True)
    def __init__(self, manifest, EOS_token='<EOS>', pad_token='<pad>', OOV_token='<OOV>', training_GPUs=(0,),
                 TARGETS_ARE_SEQUENCES=True, VERBOSE=True, layer_sizes=None, FF_dropout=None, RNN_dropout=None,
                 TEMPORALLY_CONVOLVE=None, EMA_decay=None, beam_width=None, assessment_epoch_interval=None,
                 tf_summaries_dir=None, N_epochs=None, temperature=None, N_cases=256, learning_rate=5e-4,
                 max_hyp_length=20, seed=0, assessment_GPU=0, checkpoint_path='./model.ckpt', inputs_to_occlude=None,
                 process_group=None, encoder_strides=None, input_staging='auto', engine_options=None):
        # engine_options: overrides of Seq2SeqEngine.OPTIONS (schedule diagnostics; the defaults are the product)
        # input_staging: how a fit keeps its training partitions in HBM -- 'fp32' as padded [n][T][C] arrays (a step reads 4 B per
        # input sample and packs them itself), 'bf16' as the bf16 im2row rows of the temporal convolution, made once per fit
        # (Seq2SeqEngine.pack_inputs: 2 B per sample, same bits as the per-step pack), 'auto' = bf16 for HBM-sized batches
        # (>= 256 MiB of fp32 per batch: BASELINE.json's 1024-electrode x 2000-sample configuration), fp32 otherwise
        self._engine = None
        self._engine_key = None
        self._epoch = 0
        assert 1 <= int(self.beam_width or 1) <= 16, 'beam_width (mocha-1_word_sequence.yaml:31) must lie in 1..16'

    def vprint(self, *a, **k):
        if self.VERBOSE:
            print(*a, **k)

    # ------------------------------------------------------------------ engine construction
    def _spec_from(self, subjects):
        from .engine import NetSpec
        ls = self.layer_sizes
        last = subjects[-1]
        dms = last.data_manifests
        # every 'encoder_<k>_targets' data key is an auxiliary head on encoder layer k (trainers.py:94-102, 786-799); the
        # first one runs on the overlapped schedule, further ones on the main stream
        aux_keys = sorted((k for k in dms if re.fullmatch(r'encoder_\d+_targets', k) and dms[k].num_features),
                          key=lambda k: int(k.split('_')[1]))
        self._aux_keys = aux_keys
        kw = dict(aux_layer=None)
        extra = []
        for n_, k in enumerate(aux_keys):
            layer = int(k.split('_')[1])
            dm = dms[k]
            dist = 'Gaussian' if dm.distribution == 'Gaussian' else 'categorical'
            hidden = list(ls.get('encoder_%d_projection' % layer, []))
            if n_ == 0:
                kw.update(aux_layer=layer, aux_hidden=hidden, aux_dim=int(dm.num_features), aux_dist=dist, aux_scale=float(dm.penalty_scale))
            else:
                extra.append(dict(layer=layer, hidden=hidden, dim=int(dm.num_features), dist=dist, scale=float(dm.penalty_scale)))
        kw['aux_extra'] = extra
        N = {int(s.decimation_factor) if self.TEMPORALLY_CONVOLVE else 1 for s in subjects}
        assert len(N) == 1, 'all subjects must share one decimation factor'
        # several entries in layer_sizes['encoder_embedding'] = a stack of strided conv layers whose strides multiply to the
        # decimation factor (trainers.py:406-407, 535-541).  How the reference divides the factor over the layers is in the absent
        # `machine_learning` package, so the split is taken from `encoder_strides` (ctor argument / manifest key; a restored
        # checkpoint supplies it from its weight shapes, update_net_from_saved_model) -- [BUILD-DEFINES] default when none is
        # given: the bottom layer takes the whole factor, the layers above have stride (= width) 1
        emb = list(ls['encoder_embedding'])
        Ntot = next(iter(N))
        if len(emb) > 1:
            assert self.TEMPORALLY_CONVOLVE, 'a stack of embedding layers is a temporal-convolution stack'
            strides = [int(x) for x in (self.encoder_strides or [Ntot] + [1] * (len(emb) - 1))]
            assert len(strides) == len(emb) and int(np.prod(strides)) == Ntot, \
                'encoder_strides %r must have one entry per embedding layer and multiply to the decimation factor %d' % (strides, Ntot)
            kw['conv_pre'] = [dict(out=int(o), stride=int(n)) for o, n in zip(emb[:-1], strides[:-1])]
        assert len(ls['decoder_embedding']) == 1 and len(ls['decoder_rnn']) == 1, 'one decoder embedding / RNN layer'
        return NetSpec(channels={s.subnet_id: int(s.data_manifests['encoder_inputs'].num_features) for s in subjects},
                       decimation=N.pop(), enc_embed=emb[-1], enc_rnn=list(ls['encoder_rnn']),
                       dec_embed=ls['decoder_embedding'][0], dec_rnn=ls['decoder_rnn'][0],
                       dec_proj_hidden=list(ls.get('decoder_projection', [])), vocab=int(dms['decoder_targets'].num_features),
                       dec_scale=float(dms['decoder_targets'].penalty_scale), ff_dropout=float(self.FF_dropout),
                       rnn_dropout=float(self.RNN_dropout), **kw)

    def _get_engine(self, subjects, all_subject_ids=None):
        from .engine import Seq2SeqEngine
        spec = self._spec_from(subjects)
        key = repr(spec)
        if self._engine is None or self._engine_key != key:
            dev = 'cuda:%d' % (self.training_GPUs[0] if self.training_GPUs else 0)
            # (dropout masks differ between the ranks of a data-parallel fit; the initial weights do not: fit() broadcasts them)
            self._engine = Seq2SeqEngine(spec, device=dev, seed=self.seed + 7919 * int(os.environ.get('RANK', '0')),
                                         lr=self.learning_rate, ema_decay=self.EMA_decay or 0.0, options=self.engine_options)
            self._engine.init_params(self.seed)
            self._engine_key = key
        return self._engine

    # ------------------------------------------------------------------ data staging
    def _stage(self, subject, partition):
        """All utterances of a partition as padded host arrays (static shapes => one hipGraph)."""
        ex = load_examples(subject, subject.block_ids[partition])
        if not ex:
            return None
        N = self._engine.spec.decimation
        T = max(e['encoder_inputs'].shape[0] for e in ex)
        T = -(-T // N) * N
        L = min(max(len(e['decoder_targets']) for e in ex), self.max_hyp_length)
        n = len(ex)
        C = ex[0]['encoder_inputs'].shape[1]
        X = np.zeros((n, T, C), np.float32)
        Y = np.zeros((n, L), np.int32)
        aux_keys = [k for k in getattr(self, '_aux_keys', []) if k in ex[0]] if self._engine.aux is not None else []
        As = []
        for k in aux_keys:
            a0 = np.asarray(ex[0][k])
            As.append(np.zeros((n, T) + a0.shape[1:], np.float32 if a0.dtype.kind == 'f' else np.int32))
        for i, e in enumerate(ex):
            x = e['encoder_inputs']
            X[i, :x.shape[0]] = x
            y = np.asarray(e['decoder_targets'])[:L]
            Y[i, :len(y)] = y
            for k, A_ in zip(aux_keys, As):
                a = np.asarray(e[k])[:T]
                A_[i, :a.shape[0]] = a if A_.ndim == 3 else a.reshape(-1)
        # per-utterance loss-normalisation counts (what the kernels count on the device): non-pad target tokens, and
        # ceil(valid auxiliary-target length / decimation) samples -- data parallel: every rank can sum them over a
        # GLOBAL batch without any exchange
        from .engine import Seq2SeqEngine
        Seq2SeqEngine.check_end_padded(X)
        tok = (Y != 0).sum(1).astype(np.int64)
        vals = []
        for A_ in As:
            tl = (A_ != 0).sum(1) if A_.ndim == 2 else (np.abs(A_).max(axis=2) > 0).sum(1)
            vals.append(-(-tl // N))
        A = As[0] if As else None
        # samples per utterance up to the last non-zero row -- the length the device reads off the end padding -- for the
        # length-balanced sharding of data-parallel fits (one utterance at a time: no fp32 copy of the partition)
        xlen = np.zeros(n, np.int64)
        for i in range(n):
            nz = np.flatnonzero(X[i].any(axis=1))
            xlen[i] = nz[-1] + 1 if nz.size else 0
        return dict(X=X, Y=Y, A=A, Ax=As[1:], n=n, T=T, L=L, tok=tok, val=(vals[0] if vals else np.zeros(n, np.int64)), valx=vals[1:], xlen=xlen)

    def _batches(self, data, rng=None):
        B = self.N_cases
        order = np.arange(data['n']) if rng is None else rng.permutation(data['n'])
        for i in range(0, data['n'], B):
            yield order[i:i + B]

    def _resident(self, eng, data):
        """The partition's padded arrays as device tensors, kept in HBM for the whole fit (a MOCHA-TIMIT subject is
        ~0.5 GB of fp32 ECoG against 288 GB): a step's batch is then assembled by e2t_gather_rows_u32 at HBM speed
        instead of a 105-MB pageable host->device copy per step.  Partitions beyond E2T_RESIDENT_GB (default 64) stay on
        the host and go through a pinned staging buffer."""
        import torch
        if 'dev' not in data:
            nbytes = sum(a.nbytes for a in [data['X'], data['Y'], data['A']] + list(data.get('Ax', [])) if a is not None)
            if nbytes > float(os.environ.get('E2T_RESIDENT_GB', '64')) * 2 ** 30:
                data['dev'] = None
            else:
                data['dev'] = {k: torch.from_numpy(data[k]).to(eng.device) for k in ('X', 'Y', 'A') if data[k] is not None}
                data['dev']['Ax'] = [torch.from_numpy(a).to(eng.device) for a in data.get('Ax', [])]
        return data['dev']

    def _prefetch_sources(self, eng, ws, data):
        """Hand the engine the resident arrays of this partition as the sources of its in-graph batch assembly (False: the
        partition is not resident, or the engine does not prefetch)."""
        dev = self._resident(eng, data)
        if dev is None:
            eng.set_prefetch(ws, None)
            return False
        pairs = [(dev['Y'], ws['Y']), (dev['X'], ws['X'])]
        if 'A' in dev and eng.aux is not None:
            pairs.append((dev['A'], ws['auxT']))
        return eng.set_prefetch(ws, pairs)

    def _pack_partition(self, eng, sid, data):
        """bf16-staged inputs of a training partition (input_staging), made once per fit from the resident fp32 array; None
        where the fp32 form stays (small batches, a conv stack, a partition too large to keep resident)."""
        if data is None or 'packed' in data:
            return None if data is None else data['packed']
        data['packed'] = None
        mode = self.input_staging or 'auto'
        assert mode in ('auto', 'fp32', 'bf16'), 'input_staging must be auto, fp32 or bf16'
        big = self.N_cases * data['T'] * data['X'].shape[2] * 4 >= (1 << 28)
        if mode == 'fp32' or (mode == 'auto' and not big) or not eng.packed_inputs_ok(sid):
            return None
        dev = self._resident(eng, data)
        if dev is not None:
            # the packed rows are kept BESIDE the resident fp32 partition (the assessments read x): 1.5x its bytes in all, which
            # must fit the residency budget too; a partition that does not, or an allocation the device refuses, stays fp32-staged
            import torch
            pa_bytes = data['X'].shape[0] * (-(-data['T'] // eng.spec.decimation)) * (eng.spec.decimation * data['X'].shape[2] + 64) * 2
            if data['X'].nbytes + pa_bytes <= float(os.environ.get('E2T_RESIDENT_GB', '64')) * 2 ** 30:
                try:
                    data['packed'] = eng.pack_inputs(sid, dev['X'])
                except torch.OutOfMemoryError:
                    data['packed'] = None
                    torch.cuda.empty_cache()
        return data['packed']

    def _load_batch(self, eng, ws, data, idx, idx_dev=None, packed=None):
        """Batch rows `idx` (host index array; -1 = padding utterance) into the workspace.  idx_dev: the same indices
        already on the device (an epoch's plan is uploaded once), so the step issues no host->device copy at all.
        packed: the partition's bf16-staged inputs (_pack_partition) -- the conv operand is gathered instead of x."""
        import torch
        from .hip_lib import lib
        B = ws['B']
        dev = self._resident(eng, data)
        ws['packed'] = False
        ws['have'] = None                    # (whatever batch a captured step prefetched into this workspace is gone)
        if dev is not None:
            if idx_dev is None:
                full = np.full(B, -1, np.int32)
                full[:len(idx)] = idx
                idx_dev = torch.from_numpy(full).to(eng.device)
            st = eng.stream
            pairs = [(dev['Y'], ws['Y'])]
            if packed is not None:
                eng.load_packed_batch(ws, packed, idx_dev)
            else:
                pairs.append((dev['X'], ws['X']))
            if 'A' in dev and eng.aux is not None:
                pairs.append((dev['A'], ws['auxT']))
            pairs += [(a, wx['T']) for a, wx in zip(dev.get('Ax', []), ws['auxx'])]
            for src, dst in pairs:
                lib.e2t_gather_rows_u32(src.data_ptr(), idx_dev.data_ptr(), B, B, src[0].numel(), dst.data_ptr(), st)
            ws['_keep'] = idx_dev            # alive until the next batch replaces it
            return
        keep = np.asarray(idx)[np.asarray(idx) >= 0]

        def put(dst, src):
            if '_pin' not in ws:
                ws['_pin'] = {}
            pin = ws['_pin'].get(dst.data_ptr())
            if pin is None:
                pin = ws['_pin'][dst.data_ptr()] = torch.zeros(dst.shape, dtype=dst.dtype).pin_memory()
            pin.zero_()
            np.take(src, keep, axis=0, out=pin.numpy()[:len(keep)])
            dst.copy_(pin, non_blocking=True)
        put(ws['X'], data['X'])
        put(ws['Y'], data['Y'])
        if data['A'] is not None and eng.aux is not None:
            put(ws['auxT'], data['A'])
        for a, wx in zip(data.get('Ax', []), ws['auxx']):
            put(wx['T'], a)
        torch.cuda.current_stream(eng.device).synchronize()      # the pinned buffers are reused by the next batch

    # ------------------------------------------------------------------ fit
    def fit(self, subjects, _restore_epoch=None, train_vars_scope=None, reuse_vars_scope=None):
        """Train for self.N_epochs epochs; assess every assessment_epoch_interval epochs on the last subject's
        'training' and 'validation' partitions (EMA weights); checkpoint at the end."""
        eng = self._get_engine(subjects)
        # the whole fit runs with the engine's own stream current (captured steps are replayed from it: engine.train_step),
        # ordered behind the caller's stream and joined back into it
        with eng.on_step_stream():
            return self._fit(eng, subjects, _restore_epoch, train_vars_scope, reuse_vars_scope)

    def _fit(self, eng, subjects, _restore_epoch, train_vars_scope, reuse_vars_scope):
        import torch
        start = 0
        if _restore_epoch:
            self._restore(eng, _restore_epoch, reuse_vars_scope)
            start = _restore_epoch
        names = self._tf_names(eng)
        if train_vars_scope:
            eng.trainable = {seg for seg, tf in names.items() if re.match(train_vars_scope, tf)}
        else:
            eng.trainable = None
        from .parallel import make_sync, global_batches, rank_slice
        sync = self._get_sync(eng)
        rank, world = (sync.rank, sync.world) if sync is not None else (0, 1)
        if sync is not None:
            sync.broadcast_([eng.store.p, eng.store.ema])
            eng.pack('p')
        staged = {s.subnet_id: {part: self._stage(s, part) for part in ('training', 'validation')} for s in subjects}
        last = subjects[-1]
        res = {p: AssessmentTuple(decoder_accuracies=[], decoder_word_error_rates=[], decoder_confusions=None, losses=[])
               for p in ('training', 'validation')}
        rng = np.random.default_rng(self.seed + start)           # the same permutations on every rank
        interval = self.assessment_epoch_interval or self.N_epochs
        B = self.N_cases
        for epoch in range(self.N_epochs):
            if epoch % interval == 0:
                for part in res:
                    a = self._assess(eng, last, staged[last.subnet_id][part], sync)
                    res[part].decoder_accuracies.append(a.accuracy)
                    res[part].decoder_word_error_rates.append(a.word_error_rate)
                    res[part].decoder_confusions = a.decoder_confusions
                    res[part].hypotheses, res[part].references = a.hypotheses, a.references
                self.vprint('epoch %4d  train acc %.3f WER %.3f | valid acc %.3f WER %.3f' % (
                    start + epoch, res['training'].decoder_accuracies[-1], res['training'].decoder_word_error_rates[-1],
                    res['validation'].decoder_accuracies[-1], res['validation'].decoder_word_error_rates[-1]))
            # the epoch's plan: per subject, the global batches (world x N_cases utterances), this rank's slice of each as
            # a row of device-resident indices (-1 = padding utterance) and the global loss-normalisation counts.  Every
            # rank runs every step -- also one whose slice is short or empty -- so the exchange always matches up.
            plans = []
            for s in subjects:
                d = staged[s.subnet_id]['training']
                if d is None or d['n'] == 0:
                    continue
                gb = global_batches(d['n'], B, world, rng)
                idx = np.full((len(gb), B), -1, np.int32)
                cnt = np.zeros((len(gb), 2 + len(d.get('valx', []))), np.int64)
                for k, g in enumerate(gb):
                    mine = rank_slice(g, B, rank, world, d.get('xlen') if world > 1 else None)
                    idx[k, :len(mine)] = mine
                    cnt[k] = [d['tok'][g].sum(), d['val'][g].sum()] + [v[g].sum() for v in d.get('valx', [])]
                plans.append((s.subnet_id, d, idx, torch.from_numpy(idx).to(eng.device), cnt, self._pack_partition(eng, s.subnet_id, d)))
            ws = None
            self._epoch_serial = getattr(self, '_epoch_serial', 0) + 1        # (names an epoch's plan: see `have` below)
            for k in range(max((len(p[2]) for p in plans), default=0)):      # round-robin over subjects ('parallel' learning)
                for sid, d, idx, idx_dev, cnt, pk in plans:
                    if k >= len(idx):
                        continue
                    ws = eng.workspace(sid, B, d['T'], d['L'])
                    # Round 6: with the partition resident in HBM the captured step assembles the NEXT batch itself, on a side
                    # branch under its encoder (Seq2SeqEngine.set_prefetch: 42 us per cfg2 step between two replays otherwise).
                    # `have` names the batch the workspace holds: the fit gathers only when that is not the one this step wants
                    # (first step of an epoch, an assessment in between, an eager fallback step).
                    pf = pk is None and self._prefetch_sources(eng, ws, d)
                    want = (self._epoch_serial, sid, k)
                    if not (pf and ws.get('have') == want):
                        self._load_batch(eng, ws, d, idx[k], idx_dev[k], packed=pk)
                    if pf:
                        nk = min(k + 1, len(idx) - 1)
                        ws['next_idx'].copy_(idx_dev[nk])
                    if sync is not None:
                        eng.set_global_counts(ws, int(cnt[k, 0]), int(cnt[k, 1]), [int(v) for v in cnt[k, 2:]])
                    eng.train_step(ws, sync=sync)
                    ws['have'] = (self._epoch_serial, sid, nk) if (pf and ws.get('prefetched')) else None
            if ws is not None:
                lo = eng.losses(ws)                                  # (also raises if an in-kernel wait timed out this epoch)
                if sync is not None:
                    # every rank holds its share of the globally normalised sums: the global losses are their sum
                    keys = sorted(lo)
                    lo = dict(zip(keys, sync.allreduce_numpy(np.array([lo[k] for k in keys], np.float32)).tolist()))
                res['training'].losses.append(lo)
                nsat = eng.saturation_events()
                if nsat:
                    print('WARNING: epoch %d: the persistent BPTT clipped recurrent gate gradients at |x| >= 2 in %d publishes '
                          '(check the penalty scales; engine_options=dict(persistent="fwd") selects the unclipped launch-per-step BPTT)' % (start + epoch, nsat))
        eng.check_sync()                                             # never checkpoint the results of an invalid step
        self._epoch = start + self.N_epochs
        self._save(eng, self._epoch)
        for part in res:
            res[part].decoder_accuracies = np.array(res[part].decoder_accuracies)
            res[part].decoder_word_error_rates = np.array(res[part].decoder_word_error_rates)
        return res

    def _get_sync(self, eng):
        """The gradient exchange of this process layout (None for one process); made once per engine.  A new engine (the
        subject set changed: sequential_transfer_learn fits one subject at a time) gets a new exchange; the previous
        communicator is destroyed first."""
        if getattr(self, '_sync_for', None) is not eng:
            from .parallel import make_sync
            old = getattr(self, '_sync', None)
            if old is not None and hasattr(old, 'close'):
                old.close()
            self._sync = make_sync(eng.store.g, self.process_group)
            self._sync_for = eng
        return self._sync

    # ------------------------------------------------------------------ assessment (row a11)
    def _assess(self, eng, subject, data, sync=None):
        """Token accuracy (teacher forced), greedy hypotheses, WER, confusions on one partition with the EMA weights.
        Data parallel: the utterances are sharded like the training batches, every rank decodes its slice,
        the token ids / counts are summed over the ranks and every rank computes the same metrics."""
        import torch
        from .parallel import global_batches, rank_slice
        out = AssessmentTuple()
        if data is None:
            out.accuracy = out.word_error_rate = float('nan')
            return out
        feats = list(subject.data_manifests['decoder_targets'].get_feature_list())
        V = len(feats)
        rank, world = (sync.rank, sync.world) if sync is not None else (0, 1)
        B, n, L = self.N_cases, data['n'], data['L']
        ws = eng.workspace(subject.subnet_id, B, data['T'], L)
        hyp_all = np.zeros((n, L), np.int32)
        counts = np.zeros(2, np.int64)                       # correct tokens, tokens
        conf = np.zeros((V, V), np.int64) if V < 100 else None
        for g in global_batches(n, B, world):
            idx = rank_slice(g, B, rank, world, data.get('xlen') if world > 1 else None)
            if len(idx) == 0:
                continue
            self._load_batch(eng, ws, data, idx)
            if eng._packed != 'ema':
                eng.pack('ema')
            eng.forward(ws, train=False, which='ema', with_aux=False)
            torch.cuda.synchronize(eng.device)
            nt = int(ws['ntok'].item())
            counts += (int(round(float(ws['loss'][2].item()) * max(nt, 1))), nt)
            if conf is not None:
                pred = ws['pred'].cpu().numpy().reshape(L, -1)[:, :len(idx)].T
                for b, i in enumerate(idx):
                    y = data['Y'][i]
                    for l in range(int((y != 0).sum())):
                        conf[y[l], pred[b, l]] += 1
            W = int(self.beam_width or 1)
            if W > 1:       # beam search (beam_width, temperature: mocha-1_word_sequence.yaml:31, 82); width 1 = greedy
                hyp_dev, _ = eng.beam_decode(ws, W, float(self.temperature or 1.0), which='ema')
            else:
                hyp_dev = eng.greedy_decode(ws, which='ema')
            hyp_all[idx] = hyp_dev.cpu().numpy()[:len(idx)]
        eng.check_sync(ws)                                   # a timed-out forward pass must not be reported as a result
        eng.pack('p')
        if sync is not None:
            hyp_all = sync.allreduce_numpy(hyp_all)
            counts = sync.allreduce_numpy(counts)
            if conf is not None:
                conf = sync.allreduce_numpy(conf)
        hyps = target_inds_to_sequences(hyp_all, feats)
        refs = target_inds_to_sequences(data['Y'], feats)
        out.accuracy = float(counts[0]) / max(int(counts[1]), 1)
        out.word_error_rate = float(np.mean(wer_vector(refs, hyps))) if refs else float('nan')
        out.decoder_confusions, out.hypotheses, out.references = conf, hyps, refs
        return out

    def restore_and_assess(self, subjects, restore_epoch, WRITE=False):
        eng = self._get_engine(subjects)
        self._restore(eng, restore_epoch, None)
        last = subjects[-1]
        sync = self._get_sync(eng)
        return {part: self._assess(eng, last, self._stage(last, part), sync) for part in ('training', 'validation')}

    def restore_and_get_saliencies(self, subjects, restore_epoch, data_partition='validation', assessment_type='norms'):
        """Back-propagate the (penalty-weighted) loss into the inputs (reference trainers.py:703-732).
        'norms' -> per-electrode RMS gradient [C]; 'sequences' -> [examples, T, C]."""
        import torch
        eng = self._get_engine(subjects)
        self._restore(eng, restore_epoch, None)
        subject = subjects[-1]
        data = self._stage(subject, data_partition)
        ws = eng.workspace(subject.subnet_id, self.N_cases, data['T'], data['L'])
        eng.pack('ema')
        chunks = []
        for idx in self._batches(data):
            self._load_batch(eng, ws, data, idx)
            eng.forward(ws, train=False, which='ema')
            eng.backward(ws, train=False)
            g = eng.input_gradient(ws)
            torch.cuda.synchronize(eng.device)
            chunks.append(g[:len(idx)].cpu().numpy())
        eng.pack('p')
        G = np.concatenate(chunks, 0)
        if assessment_type == 'sequences':
            return G
        return np.sqrt((G ** 2).mean(axis=(0, 1)))

    def restore_and_get_internal_activations(self, subjects, restore_epoch, data_partition='validation'):
        """The tensors the reference's activation probe fetches (trainers.py:757-765) for the last subject:
        'reversed_inputs' [n,T,C] (time-reversed over each utterance's valid length, trainers.py:808-810),
        'convolved_inputs' [n,S,F] (front-end output, S = T/decimation), 'decimated_reversed_targets' [n,S,K]
        (auxiliary encoder targets, reversed then every decimation-th sample, trainers.py:791-795; None without
        an auxiliary target) and 'final_RNN_state' [2,n,2H] = (c, h) of the top encoder layer at each utterance's own
        last step (what initialises the decoder).  Computed with the EMA weights, dropout off."""
        import torch
        eng = self._get_engine(subjects)
        self._restore(eng, restore_epoch, None)
        subject = subjects[-1]
        data = self._stage(subject, data_partition)
        ws = eng.workspace(subject.subnet_id, self.N_cases, data['T'], data['L'])
        s = eng.spec
        S, B, F = ws['S'], ws['B'], s.enc_embed
        out = dict(reversed_inputs=[], convolved_inputs=[], decimated_reversed_targets=[], final_RNN_state=[])
        for idx in self._batches(data):
            self._load_batch(eng, ws, data, idx)
            if eng._packed != 'ema':
                eng.pack('ema')
            eng.forward(ws, train=False, which='ema')
            torch.cuda.synchronize(eng.device)
            n = len(idx)
            lens = ws['lens'].cpu().numpy()[:n]
            X = data['X'][idx]
            R = np.zeros_like(X)
            for i, ln in enumerate(lens):
                R[i, :ln] = X[i, :ln][::-1]
            out['reversed_inputs'].append(R)
            E = ws['E'].float().cpu().numpy().reshape(S, B, -1)[:, :n, :F]
            out['convolved_inputs'].append(E.transpose(1, 0, 2))
            if ws.get('use_aux'):
                At = ws['At'].cpu().numpy().reshape(S, B, -1)[:, :n]
                out['decimated_reversed_targets'].append(At.transpose(1, 0, 2))
            H2 = 2 * s.enc_rnn[-1]
            h = ws['dec']['Yext'][:B].float().cpu().numpy()[:n, :H2]
            c = ws['c0'].cpu().numpy()[:n, :H2]
            out['final_RNN_state'].append(np.stack([c, h]))
        eng.pack('p')
        res = {k: (np.concatenate(v, axis=1 if k == 'final_RNN_state' else 0) if v else None) for k, v in out.items()}
        return res

    def online_predictor(self, subjects, restore_epoch, targets_list=None, subject=None, max_length=None):
        """predict(inputs) for one utterance [T,C] or a batch [B,T,C] of (zero-padded) ECoG: greedy word sequences
        from the restored EMA weights -- the device-resident counterpart of the reference's SavedModel predictor
        (trainers.py:925-949).  With targets_list the hypotheses come back as strings, else as token-id arrays."""
        import torch
        eng = self._get_engine(subjects)
        self._restore(eng, restore_epoch, None)
        subject = subject or subjects[-1]
        N = eng.spec.decimation
        L = int(max_length or self.max_hyp_length)
        eng.pack('ema')

        def predict(inputs):
            x = np.asarray(inputs, np.float32)
            x = x[None] if x.ndim == 2 else x
            B, T = x.shape[0], -(-x.shape[1] // N) * N
            ws = eng.workspace(subject.subnet_id, B, T, L)
            ws['packed'] = False
            ws['X'].zero_()
            ws['X'][:, :x.shape[1]].copy_(torch.from_numpy(np.ascontiguousarray(x)))
            ws['Y'].zero_()
            hyp = eng.greedy_decode(ws, which='ema').cpu().numpy()
            eng.check_sync(ws)
            return target_inds_to_sequences(hyp, list(targets_list)) if targets_list is not None else hyp
        return predict

    # ------------------------------------------------------------------ checkpoints (rows a13)
    def _tf_names(self, eng):
        """segment name -> representative TF-style variable name (for the scope regexes of trainers.py:337-366)."""
        out = {}
        for seg in eng.store.order:
            if seg.startswith('conv'):
                out[seg] = 'seq2seq/subnet_%s/encoder_embedding' % re.fullmatch(r'conv(.*)\.W\d*', seg).group(1)
            elif seg.startswith('enc'):
                out[seg] = 'seq2seq/encoder_rnn_%s' % seg[3:].split('.')[0]
            elif seg.startswith('auxx'):
                out[seg] = 'seq2seq/encoder_%s_projection' % eng.spec.aux_extra[int(seg[4:].split('_')[0])]['layer']
            elif seg.startswith('aux'):
                out[seg] = 'seq2seq/encoder_%s_projection' % eng.spec.aux_layer
            elif seg == 'dec.emb':
                out[seg] = 'seq2seq/decoder_embedding'
            elif seg.startswith('dec.'):
                out[seg] = 'seq2seq/decoder_rnn'
            else:
                out[seg] = 'seq2seq/decoder_projection'
        return out

    def _ckpt(self, epoch):
        return '%s-%d' % (self.checkpoint_path, epoch)

    def _save(self, eng, epoch):
        import torch
        sync = getattr(self, '_sync', None)
        if (sync.rank if sync is not None else int(os.environ.get('RANK', '0'))) != 0:
            # the other ranks wait until rank 0 has the files in place: the next fit(_restore_epoch=...) / restore_and_assess
            # restores on EVERY rank and must not meet a missing or half-written checkpoint
            if sync is not None:
                sync.barrier()
            return
        # float32, as the reference's TF1 Saver stores (and restores into) its variables: a DT_DOUBLE entry would be
        # rejected by the reference on restore
        try:
            arrays = {k: v.astype(np.float32) for k, v in eng.store.export_tf('p').items()}
            arrays.update({k + EMA_SUFFIX: v.astype(np.float32) for k, v in eng.store.export_tf('ema').items()})
            arrays['__adam_m'] = eng.store.m.cpu().numpy()
            arrays['__adam_v'] = eng.store.v.cpu().numpy()
            arrays['__step'] = eng.step_t.cpu().numpy()
            ckdir = os.path.dirname(self.checkpoint_path) or '.'
            os.makedirs(ckdir, exist_ok=True)
            from . import tf_checkpoint
            for f in os.listdir(ckdir):                        # temporaries a crashed writer left behind -- not those of a LIVE
                m = re.match(r'\.tmp-(\d+)-(?:h([0-9a-f]{8})-)?', f)     # process writing into the same directory (another fit or assessment)
                if not m:
                    continue
                if m.group(2) is not None and m.group(2) != tf_checkpoint.host_tag():
                    # another HOST's writer (shared file system): its process id says nothing here -- reaped by age only
                    try:
                        stale = time.time() - os.path.getmtime(os.path.join(ckdir, f)) > 3600.0
                    except OSError:
                        continue
                else:
                    stale = int(m.group(1)) == os.getpid() or not _pid_alive(int(m.group(1)))
                if stale:
                    try:
                        os.remove(os.path.join(ckdir, f))
                    except OSError:
                        pass
            # written under temporary names and moved into place (os.replace is atomic): a reader never sees a partial file;
            # the names start with '.tmp-', which the trainer's restore scan ('model.ckpt-<epoch>.index') cannot match
            final = self._ckpt(epoch)
            tmp = os.path.join(ckdir, tf_checkpoint.temp_prefix() + os.path.basename(final))
            np.savez(tmp + '.npz', **arrays)
            os.replace(tmp + '.npz', final + '.npz')
            # the same variables (weights + EMA shadows, reference naming grammar) as a TensorFlow V2 checkpoint:
            # `model.ckpt-<epoch>.index` is what the trainer's restore_epoch scan looks for (trainers.py:235-252), and the
            # pair is readable by TF's own checkpoint reader (recover_model_sizes, trainers.py:444-554)
            tf_checkpoint.write_checkpoint(final, {k: v for k, v in arrays.items() if not k.startswith('__')})
        finally:
            # (the barrier is reached whatever happened above: the other ranks are waiting in theirs; the error is re-raised)
            if sync is not None:
                sync.barrier()

    def _restore(self, eng, epoch, reuse_vars_scope):
        import torch
        z = load_checkpoint_arrays(self._ckpt(epoch))
        P = {k: z[k] for k in z.files if not k.startswith('__') and not k.endswith(EMA_SUFFIX)}
        E = {k[:-len(EMA_SUFFIX)]: z[k] for k in z.files if k.endswith(EMA_SUFFIX)}
        if reuse_vars_scope is not None:
            # variables outside the reused scope keep their fresh initialisation (trainers.py:352-353)
            cur_p, cur_e = eng.store.export_tf('p'), eng.store.export_tf('ema')
            for k in list(P):
                if not re.match(reuse_vars_scope, k) or k not in cur_p or cur_p[k].shape != P[k].shape:
                    P[k], E[k] = cur_p.get(k, P[k]), cur_e.get(k, E[k])
        have = eng.store.export_tf('p')
        P = {k: P.get(k, have[k]) for k in have}
        E = {k: E.get(k, P[k]) for k in have}
        eng.store.import_tf(P, bufs=('p',))
        eng.store.import_tf(E, bufs=('ema',))
        if '__adam_m' in z.files and z['__adam_m'].shape[0] == eng.store.n and reuse_vars_scope in (None, 'seq2seq'):
            eng.store.m.copy_(torch.from_numpy(z['__adam_m']))
            eng.store.v.copy_(torch.from_numpy(z['__adam_v']))
            eng.step_t.copy_(torch.from_numpy(z['__step']))
        eng.pack('p')

    def get_weights_as_numpy_array(self, full_var_name, epoch):
        z = load_checkpoint_arrays(self._ckpt(epoch), names=[full_var_name])
        return z[full_var_name]
