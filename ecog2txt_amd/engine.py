"""Device engine of the ECoG->text hot path: parameter store, operand packing,
workspaces and the launch sequences for train / assess / greedy decode.

This is host-side plumbing only: every FLOP runs in libecog2txt_hip.so
(hand-written gfx950 kernels) through the C ABI of include/ecog2txt_hip.h.
torch supplies device memory, streams, hipGraph capture and torch.distributed.
There is no CPU fallback: constructing an engine without a GPU or without the
built library raises.

Reference stages replaced (all inside the un-vendored SequenceNetwork.fit,
ecog2txt/trainers.py:318): reverse inputs (trainers.py:808-810) ->
_convolve_sequences (813-818) -> _encode_sequences (821-823) ->
_prepare_encoder_targets (798-799) -> decoder + losses -> Adam/EMA.
Parameter naming follows the checkpoint grammar recovered by
MultiSubjectTrainer.recover_model_sizes (trainers.py:444-554).
"""
import contextlib
import ctypes as C
import re

import numpy as np
import torch

from . import hip_lib as H
from .hip_lib import lib
# (NetSpec, ParamStore and the layout helpers are part of this module's interface: callers import them from here)
from .params import (NetSpec, ParamStore, EOS_ID, PAD_ID, OOV_ID, STREAM_AUX, STREAM_CONV, STREAM_CONV_PRE, STREAM_DEC_EMB,   # noqa: F401
                     STREAM_DEC_OUT, STREAM_ENC, capture, ceil_div, conv_seg, conv_stack, conv_tf_name, r8, rk)
from .layers import _FFStack, _Lstm, _bf, _f32, _i32
from .packing import PackingMixin
from .decoding import DecodingMixin


class Seq2SeqEngine(PackingMixin, DecodingMixin):
    # Schedule options (constructor argument `options`, a dict of overrides; diagnostics and tests -- the defaults are the product):
    #   persistent   '1' whole-sequence weight-stationary recurrences / '0' one launch per time step / 'fwd', 'bwd' one side only
    #   overlap      weight gradients, optimiser and re-pack on side branches of the captured step (False: one stream)
    #   fused_conv   'auto' one-pass front-end (e2t_conv_fwd_fused) for HBM-sized batches (>= 256 MiB) / '1' always / '0' never
    #   tn           weight gradients straight from the K-major activations (False: operand transposes + K-contiguous GEMM)
    #   group_gemms  the K-major products of a backward stage in one grouped launch (False: one launch per product)
    #   launch_stream  captured steps replayed from a stream of the engine's own (False: the caller's stream)
    #   fused_tail     the captured step's early optimiser update and the re-pack of its images as ONE pass over the weight matrices
    #                  (e2t_adam_pack_batch; False: e2t_adam_ema_step, then e2t_pack_batch -- the same bits)
    #   fused_reduce   (with fused_tail, single GPU) the weight gradients of the early ranges leave their split-K slabs un-reduced and the
    #                  fused kernel sums them as it reads the gradient (E2T_GEMM_KEEP_SLABS; False: reduction launches as ever)
    #   dp_one_graph   data parallel: the step as ONE graph with the collectives as nodes; False (the default until that schedule has
    #                  run with more than one RCCL rank): one graph per backward stage, the collectives issued eagerly between them --
    #                  also the fallback ALL ranks take together when any rank's capture is refused
    #   fused_bottom   (with fused_tail, single GPU) the BOTTOM encoder layer's update at the end of the captured step goes through the fused
    #                  kernel too (slab sums + Adam + EMA + its operand images in one pass) instead of reduction + e2t_adam_ema_step now and
    #                  e2t_pack_batch at the start of the next step
    #   prefetch_batches  (round 6 experiment, OFF) a fit whose partitions are resident in HBM: the captured step gathers the NEXT batch into its
    #                  own input buffers on a side branch (set_prefetch); False: the fit gathers between steps (42 us per cfg2 step).  Same
    #                  results (tests/test_gpu_e2e.py), but SLOWER wherever the gather was placed: 1.78 vs 1.68 ms per step at fit level
    #   lean_critical  the decoder's loss and accuracy sums in one launch (e2t_sum2_f32), the zero fill of the embedding gradient on the
    #                  weight-gradient branch that uses it: two launches fewer on the critical branch between forward and backward pass
    #   small_batch_head  greedy decoding of <= 8 utterances: one head launch per token (e2t_greedy_head_small) instead of gather + GEMM + arg-max
    #   big_bptt_masks  a large layer (lstm_big) applies its output-dropout mask to dY inside its BPTT, so the producers of dY (the input
    #                   gradient of the layer above: the 256 x 256 lean-epilogue instance then takes it) do not (layers._Lstm.out_drop).
    #                   Round 6, measured and left OFF: cfg4 8.27 / 8.21 ms with it against 8.17 / 8.19 without (two same-box pairs)
    OPTIONS = dict(persistent='1', overlap=True, fused_conv='auto', tn=True, group_gemms=True, launch_stream=True, dp_one_graph=False, fused_tail=True, fused_reduce=True,
                   big_bptt_masks=False, fused_bottom=True, small_batch_head=True, lean_critical=True, prefetch_batches=False)

    def __init__(self, spec, device='cuda:0', seed=0, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, ema_decay=0.99, options=None):
        if not torch.cuda.is_available():
            raise RuntimeError('ecog2txt_amd needs an MI355X (HIP) device; there is no CPU fallback for this path')
        H.load()
        self.spec = spec
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.seed = int(seed)
        self.hyper = dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, ema_decay=ema_decay)
        self.grad_scale = 1.0
        self.store = ParamStore(spec, self.device)
        self.step_t = _i32(1, device=self.device)
        s = spec
        assert all(h % 2 == 0 for h in s.enc_rnn) and s.dec_rnn % 2 == 0, 'hidden sizes must be even'
        assert s.dec_rnn == 2 * s.enc_rnn[-1], 'decoder state = concat(fwd, bwd) encoder state (App. D2)'
        dev = self.device
        self.F8 = rk(s.enc_embed)
        # conv operand per subject: B operand [F][Kc8]; input-gradient operand [Kc8][F8] made on demand
        # conv stack per subject.  Bottom layer: B operand [out_0][rk(N_0*C + 1)] of the im2row product.  Layer j >= 1 reads the
        # layer below through a VIEW (rows in grouped order, e2t_conv_pack_grouped): operand [out_j][N_j * ld_{j-1}] with the
        # weights of tap w at columns w*ld_{j-1} .., and [N_j * ld_{j-1}][ld_j] for the input gradient.  ld of an inner layer's
        # output = rk(r8(out) + 1): the ones column (bias gradient of the layer above) sits at the 16-B aligned index r8(out)
        self.conv = {sid: conv_stack(s, Cc) for sid, Cc in s.channels.items()}
        self.conv_ld = {sid: [rk(r8(co) + 1) for (_, co, _) in lays[:-1]] + [self.F8] for sid, lays in self.conv.items()}
        self.conv_G = {sid: [int(np.prod([n for (_, _, n) in lays[j + 1:]] or [1])) for j in range(len(lays))] for sid, lays in self.conv.items()}
        assert not s.conv_pre or s.conv_relu, 'a conv stack is built with ReLU layers (the mask of the input gradient is E != 0)'
        self.convT = {sid: _bf(lays[0][1], rk(lays[0][2] * lays[0][0] + 1), device=dev) for sid, lays in self.conv.items()}
        self.convTj = {sid: {j: _bf(lays[j][1], lays[j][2] * self.conv_ld[sid][j - 1], device=dev) for j in range(1, len(lays))}
                       for sid, lays in self.conv.items()}
        self.convBj = {sid: {j: _bf(lays[j][2] * self.conv_ld[sid][j - 1], self.conv_ld[sid][j], device=dev) for j in range(1, len(lays))}
                       for sid, lays in self.conv.items()}
        self.convB = {}
        self.enc = []
        for l, Hh in enumerate(s.enc_rnn):
            if l == 0:
                D, blocks, ld = s.enc_embed, [(0, s.enc_embed, 0)], self.F8
            else:
                Hp = s.enc_rnn[l - 1]
                D, blocks, ld = 2 * Hp, [(0, Hp, 0), (Hp, Hp, r8(Hp))], rk(2 * r8(Hp) + 1)
            self.enc.append(_Lstm(self, 'enc%d' % l, 2, D, blocks, ld, Hh, STREAM_ENC + l))
        self.aux = None
        if s.aux_layer is not None:
            Hk = s.enc_rnn[s.aux_layer]
            self.aux = _FFStack(self, 'aux', [2 * Hk] + list(s.aux_hidden) + [s.aux_dim],
                                [(0, Hk, 0), (Hk, Hk, r8(Hk))], rk(2 * r8(Hk) + 1), STREAM_AUX)
        self.aux_x = []
        for j, hx in enumerate(s.aux_extra):
            assert hx['layer'] != s.aux_layer and 0 <= hx['layer'] < len(s.enc_rnn), 'one auxiliary head per encoder layer'
            Hk = s.enc_rnn[hx['layer']]
            self.aux_x.append(_FFStack(self, 'auxx%d_' % j, [2 * Hk] + list(hx.get('hidden', [])) + [hx['dim']],
                                       [(0, Hk, 0), (Hk, Hk, r8(Hk))], rk(2 * r8(Hk) + 1), STREAM_AUX + 4 * (j + 1)))
        self.E8 = rk(s.dec_embed)
        self.emb = _bf(s.vocab, self.E8, device=dev)
        self.dec = _Lstm(self, 'dec', 1, s.dec_embed, [(0, s.dec_embed, 0)], self.E8, s.dec_rnn, STREAM_DEC_OUT)
        self.proj = _FFStack(self, 'proj', [s.dec_rnn] + list(s.dec_proj_hidden) + [s.vocab],
                             [(0, s.dec_rnn, 0)], rk(r8(s.dec_rnn) + 1), STREAM_DEC_OUT + 1)
        # does the buffer a layer reads as x carry the ones column at x[:, D]?  (set where the buffers are allocated)
        self.enc[0].ones_col_set = self.F8 > s.enc_embed
        for l in range(1, len(self.enc)):
            self.enc[l].ones_col_set = self.enc[l - 1].ldy > 2 * self.enc[l - 1].H8
        self.dec.ones_col_set = self.E8 > s.dec_embed
        self.proj.ones_col_set = self.dec.ldy > self.dec.H8 and self.dec.H8 == s.dec_rnn      # the column must be x[:, fin]
        if self.aux:
            k = s.aux_layer
            self.aux.ones_col_set = self.enc[k].ldy > 2 * self.enc[k].H8
        for ax, hx in zip(self.aux_x, s.aux_extra):
            ax.ones_col_set = self.enc[hx['layer']].ldy > 2 * self.enc[hx['layer']].H8
        self._pack_table = None
        self._pack_ops, self._pack_sub = None, {}
        self._fused_plans = {}        # element ranges -> tables of e2t_adam_pack_batch (packing._fused_update_plan)
        self._keep_slabs, self._slab_log, self._arenas, self._arena_seq, self._group_arena = None, [], [], 0, None
        self._fused_sigs, self._fused_retired = {}, []
        self._img_early = None        # 'all' after a full pack of the masters, else the ranges the last replay re-packed itself
        self._gemm_log = None         # a list while bench.py records the step's products (instance, shape, flops)
        self._in_group = False
        self._group = None            # a list while gemm_group() collects a stage's K-major weight-gradient products
        opt = dict(self.OPTIONS)
        unknown = set(options or {}) - set(opt)
        assert not unknown, 'unknown engine options %r (known: %r)' % (sorted(unknown), sorted(opt))
        opt.update(options or {})
        for k, v in opt.items():              # (command lines hand over strings)
            if isinstance(self.OPTIONS[k], bool) and isinstance(v, str):
                assert v.lower() in ('0', '1', 'false', 'true'), (k, v)
                opt[k] = v.lower() in ('1', 'true')
        self.options = opt
        self.group_gemms = bool(opt['group_gemms'])
        mode = str(opt['persistent'])
        assert mode in ('0', '1', 'fwd', 'bwd')
        self.persistent = mode != '0'
        self.persistent_fwd, self.persistent_bwd = mode in ('1', 'fwd'), mode in ('1', 'bwd')
        self.num_cus = H.load().e2t_device_cus(self.device.index or 0)
        self.sync_err = _i32(16, device=dev)      # raised by a bounded in-kernel wait that gave up (persistent recurrence)
        self.chains = 1          # >1 measured slower: a step launch is bound by chip-level L2-miss traffic, not latency
        self._side = []
        self._ws = {}
        self._packed = None
        self.splitk_ws = _f32(16 * 1024 * 1024, device=dev)          # 64 MiB of split-K partial slabs
        self.splitk_ws_side = _f32(16 * 1024 * 1024, device=dev)     # ... of the side stream (weight-gradient branch)
        self._on_side = False
        self._wstream = None
        self._ustream = None          # data parallel: the early optimiser update's stream inside the captured step
        self._lstream = None
        self._dec_table = None          # decoding: the decoder's input projection of every token (decoding.py)
        self._dec_table_key, self._img_version, self._head_scratch = None, 0, None
        self.small_batch_head_max = 2   # greedy decoding: the one-launch head (e2t_greedy_head_small) up to this many utterances
        self.launch_stream_on = bool(opt['launch_stream'])
        # (measured and dropped in rounds 1-2, DESIGN.md appendix: weight gradients on TWO side streams, the BPTT chain alone
        #  on the chip with all weight gradients behind it, per-stage joins)
        self.overlap = bool(opt['overlap'])
        # front-end in ONE pass over x (e2t_conv_fwd_fused: reversal + im2row + bf16 rounding in the DMA path of the product, the
        # packed copy for the backward pass emitted on the way): 'auto' = when the input batch is HBM-sized (>= 256 MiB: cfg5
        # 894 -> 483 us inference / 729 us training; at cfg2's 105 MB the two-kernel path is as fast), '1' / '0' force it
        self.fused_conv = str(opt['fused_conv'])
        assert self.fused_conv in ('auto', '0', '1')
        self.tn = bool(opt['tn'])       # weight gradients straight from the K-major activations (no transposes)
        self.trainable = None         # None = everything; else set of segment names

    def init_params(self, seed=0):
        """Glorot-uniform weights, zero biases [BUILD-DEFINES]; masters and EMA shadows start equal."""
        gen = torch.Generator(device='cpu').manual_seed(int(seed))
        st = self.store
        st.p.zero_()
        for nm in st.order:
            off, shape = st.segs[nm]
            if nm.endswith('.b'):
                continue
            if nm.endswith('.Wh'):
                fi, fo, rows = shape[1], shape[2], None
            elif nm.endswith('.WT') or nm == 'dec.emb':
                fi, fo, rows = shape[1], shape[0], None
            else:                              # [in+1][out] with the bias as last row
                fi, fo, rows = shape[0] - 1, shape[1], shape[0] - 1
            lim = float(np.sqrt(6.0 / (fi + fo)))
            w = (torch.rand(shape, generator=gen) * 2 - 1) * lim
            if rows is not None:
                w[rows:] = 0.0
            st.view(nm).copy_(w)
        st.ema.copy_(st.p)
        st.m.zero_(); st.v.zero_(); self.step_t.zero_()
        self.pack('p')

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def gemm(self, A, lda, B, ldb, Cp, ldc, M, N, K, bias=None, relu=False, out_bf16=False, accumulate=False,
             drop=None, mask_src=None, row_lens=None, alpha=1.0, splitk=False, last_col_out=None, tn=False, batch=None, alg=None,
             row_group=1, ones_last_row=False):
        """batch = (n, a_stride, b_stride, c_stride): n products of the same shape in one launch (element strides).
        alg = (M, N, K) of the product WITHOUT layout padding, for flop accounting in the launch log (bench.py)."""
        ep = H.GemmEpilogue()
        if batch is not None:
            ep.batch, ep.a_batch_stride, ep.b_batch_stride, ep.c_batch_stride = batch
        ep.bias = bias
        ep.alpha = alpha
        ep.last_col_out = last_col_out
        flags = (H.GEMM_RELU if relu else 0) | (H.GEMM_OUT_BF16 if out_bf16 else 0) | (H.GEMM_ACCUMULATE if accumulate else 0)
        if splitk:
            flags |= H.GEMM_SPLITK
        if ones_last_row:              # (K-major products: A's column M-1 is the ones column -- row M-1 of the product = column sums of B)
            flags |= H.GEMM_LAST_ROW_ONES
        # the workspace is always offered: the library also splits K on its own when a product has too few tiles
        wsb = self.splitk_ws_side if self._on_side else self.splitk_ws
        keep_info = None
        if self._keep_slabs is not None and tn and splitk and self._keeps(Cp, M, N, ldc, batch, alpha, accumulate, last_col_out, bias):
            # the captured single-GPU step leaves a weight gradient's split-K slabs for the fused optimiser kernel to sum
            # (E2T_GEMM_KEEP_SLABS): the slabs need a workspace of their own until that kernel has run -- one arena per launch
            keep_info = H.SlabInfo()
            ep.slabs_out = C.pointer(keep_info)
            flags |= H.GEMM_KEEP_SLABS
        ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        if drop is not None and drop[0] > 0:
            flags |= H.GEMM_DROPOUT
            ep.drop_rate, ep.drop_stream, ep.drop_ld = drop[0], drop[1], drop[2]
            ep.drop_seed, ep.drop_step = self.seed, self.step_t.data_ptr()
        if mask_src is not None:
            ep.relu_bwd_src, ep.ld_relu_bwd_src = mask_src
        if row_lens is not None:
            ep.row_lens, ep.rows_per_step = row_lens
            ep.row_group = row_group
        ep.flags = flags
        to_group = self._group is not None and tn and splitk and self._plan_tile(tn, M, N, K, ep) != 256
        if keep_info is not None and not to_group:
            # a kept product that keeps a launch of its OWN (the 256 x 256 instance, or no group open) gets an arena of its own:
            # the group's arena is dealt from offset 0 by the grouped launch at the end of the stage, and two launches must never
            # share slab space before the fused optimiser kernel has read it (every arena has the workspace's size, so
            # the launch plan is the same)
            wsb = self._next_arena()
            ep.splitk_ws, ep.splitk_ws_bytes = wsb.data_ptr(), wsb.numel() * 4
        if to_group:
            # inside `with self.gemm_group():` -- the K-major weight-gradient products of a stage leave in ONE launch.  (A
            # product large enough for the 256 x 256 instance -- H = 1024: 2049 x 8192 x 8704 -- keeps its own launch: the
            # grouped kernel works on 128 x 128 tiles, half the flops per staged byte; cfg4 9.49 -> 9.19 ms.)
            if self._group_arena is not None:          # (one workspace per group: the first call's serves all its products)
                ep.splitk_ws, ep.splitk_ws_bytes = self._group_arena.data_ptr(), self._group_arena.numel() * 4
            self._group.append((A, lda, B, ldb, Cp, ldc, M, N, K, ep, alg or (M, N, K), batch[0] if batch is not None else 1, keep_info))
            return
        if self._gemm_log is not None:
            tile, splits = C.c_int(0), C.c_int(0)
            lib.e2t_gemm_plan(int(tn), M, N, K, C.byref(ep), C.byref(tile), C.byref(splits))
            am, an, ak = alg or (M, N, K)
            nb = batch[0] if batch is not None else 1
            self._gemm_log.append(dict(inst=('tn%d' % tile.value if tn else 'nt%d' % tile.value), M=M, N=N, K=K, batch=nb, splits=splits.value,
                                       flops=2 * am * an * ak * nb, out_bytes=(2 if out_bf16 else 4) * am * an * nb,
                                       in_bytes=2 * (am * ak + an * ak) * nb, side=self._on_side,
                                       call=(A, lda, B, ldb, Cp, ldc, M, N, K), ep=ep, tn=tn))
        (lib.e2t_gemm_tn_bf16 if tn else lib.e2t_gemm_nt_bf16)(A, lda, B, ldb, Cp, ldc, M, N, K, C.byref(ep), self.stream)
        if keep_info is not None:
            self._log_slabs(Cp, M, N, batch, keep_info)

    # ---- split-K slabs left for the fused optimiser kernel (E2T_GEMM_KEEP_SLABS) ----
    def _keeps(self, Cp, M, N, ldc, batch, alpha, accumulate, last_col_out, bias):
        """May this K-major split product leave its slabs un-reduced?  Its gradient must lie inside the ranges the fused kernel
        updates, be a dense fp32 [M][N] with nothing between the sum and the store, and end on the 64-float grid of the tile
        descriptors (so that every element of it is reached by a descriptor that knows about the slabs)."""
        if accumulate or alpha != 1.0 or last_col_out is not None or bias is not None or ldc != N or (M * N) % 64 or N % 4:
            return False
        off = (Cp - self.store.g.data_ptr()) // 4
        nb, cbs = (batch[0], batch[3]) if batch is not None else (1, M * N)
        if off % 64 or (nb > 1 and cbs != M * N):
            return False
        return any(a <= off and off + nb * M * N <= b for a, b in self._keep_slabs)

    @staticmethod
    def _slab_sig(log):
        return tuple(sorted((e['off'], e['M'], e['N'], e['batch'], e['slab'], e['splits'], e['stride']) for e in log))

    def _next_arena(self):
        i = self._arena_seq
        self._arena_seq += 1
        while len(self._arenas) <= i:                     # (allocated by the eager pass in front of a capture, never inside one)
            self._arenas.append(_f32(self.splitk_ws_side.numel(), device=self.device))
        return self._arenas[i]

    def _log_slabs(self, Cp, M, N, batch, info):
        if info.splits > 1:
            off = (Cp - self.store.g.data_ptr()) // 4
            self._slab_log.append(dict(off=off, M=M, N=N, batch=batch[0] if batch is not None else 1, slab=info.slab, splits=info.splits, stride=info.stride))

    def _plan_tile(self, tn, M, N, K, ep):
        tile, splits = C.c_int(0), C.c_int(0)
        lib.e2t_gemm_plan(int(tn), M, N, K, C.byref(ep), C.byref(tile), C.byref(splits))
        return tile.value

    def gemm_replay(self, rec):
        """Re-issue a logged launch (same operands, same epilogue) on the current stream."""
        if rec.get('group') is not None:
            lib.e2t_gemm_tn_group_bf16(len(rec['group']), rec['group'], self.stream)
            return
        (lib.e2t_gemm_tn_bf16 if rec['tn'] else lib.e2t_gemm_nt_bf16)(*rec['call'], C.byref(rec['ep']), self.stream)

    def gemm_group(self):
        """Context manager: the K-major split-K products (weight gradients) issued inside it are collected and launched as
        ONE grouped kernel (+ one grouped reduction) on exit -- e2t_gemm_tn_group_bf16: their workgroups share the chip's
        rounds instead of each product paying its own ramp-up, partly filled last round and launch gap."""
        eng = self

        class _G:
            def __enter__(self_g):
                assert not eng._in_group
                eng._in_group = True
                eng._group = [] if eng.group_gemms else None
                eng._group_arena = eng._next_arena() if (eng._keep_slabs is not None and eng._group is not None) else None
                return self_g

            def __exit__(self_g, et, ev, tb):
                items, eng._group = eng._group, None
                eng._in_group = False
                eng._group_arena = None
                if et is not None or not items:
                    return False
                eng._launch_group(items)
                return False
        return _G()

    def _launch_group(self, items):
        for k in range(0, len(items), 8):
            part = items[k:k + 8]
            calls = (H.GemmCall * len(part))()
            keep = []
            for c, it in zip(calls, part):
                A, lda, B, ldb, Cp, ldc, M, N, K, ep = it[:10]
                c.A, c.lda, c.B, c.ldb, c.C, c.ldc, c.M, c.N, c.K = A, lda, B, ldb, Cp, ldc, M, N, K
                c.ep = C.pointer(ep)
                keep.append(ep)
            if self._gemm_log is not None:
                fl = sum(2 * it[10][0] * it[10][1] * it[10][2] * it[11] for it in part)
                self._gemm_log.append(dict(inst='tn128g', M=part[0][6], N=part[0][7], K=part[0][8], batch=len(part), splits=0,
                                           flops=fl, out_bytes=sum(4 * it[10][0] * it[10][1] * it[11] for it in part),
                                           in_bytes=sum(2 * (it[10][0] + it[10][1]) * it[10][2] * it[11] for it in part),
                                           side=self._on_side, group=calls, keep=keep, tn=True,
                                           desc=' + '.join('%dx%dx%d%s' % (it[10][0], it[10][1], it[10][2], ('x%d' % it[11]) if it[11] > 1 else '') for it in part)))
            lib.e2t_gemm_tn_group_bf16(len(part), calls, self.stream)
            for it in part:
                if len(it) > 12 and it[12] is not None:
                    self._log_slabs(it[4], it[6], it[7], (it[11], 0, 0, it[6] * it[7]) if it[11] > 1 else None, it[12])

    def _dropout(self, rate, stream):
        d = H.Dropout()
        d.rate, d.seed, d.step, d.stream = rate, self.seed, self.step_t.data_ptr(), stream
        return d

    # ------------------------------------------------------------------ concurrent row-block chains
    def run_chains(self, B, launch):
        """A recurrence is latency-bound and its 64-utterance row blocks never interact inside a layer, so the
        S step launches are issued as up to `self.chains` independent chains on side streams (parallel branches
        of the captured hipGraph): one chain's launch + memory latency overlaps another chain's work.
        launch(rb_begin, rb_count, stream_handle)."""
        nrb = ceil_div(B, 64)
        n = max(1, min(self.chains, nrb))
        cur = torch.cuda.current_stream(self.device)
        if n == 1:
            launch(0, nrb, cur.cuda_stream)
            return
        while len(self._side) < n:
            self._side.append(torch.cuda.Stream(device=self.device))
        fork = torch.cuda.Event()
        fork.record(cur)
        base, rem = divmod(nrb, n)
        lo = 0
        for c in range(n):
            cnt = base + (1 if c < rem else 0)
            st = self._side[c]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                launch(lo, cnt, st.cuda_stream)
                done = torch.cuda.Event()
                done.record(st)
            cur.wait_event(done)
            lo += cnt

    def load_params(self, P):
        self.store.import_tf(P)
        self.pack('p')

    # ------------------------------------------------------------------ workspace
    def workspace(self, sid, B, T, L):
        key = (sid, B, T, L)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        s, dev = self.spec, self.device
        Cc, N = s.channels[sid], s.decimation
        S = ceil_div(T, N)
        M, Mk = S * B, rk(S * B)
        # conv stack (one layer unless spec.conv_pre): layer 0 is the im2row product over x, rows in grouped order when layers
        # follow (G0 = product of their strides), so that every later layer reads the one below through a view
        lays, lds, Gs = self.conv[sid], self.conv_ld[sid], self.conv_G[sid]
        N0, out0, ld0, G0 = lays[0][2], lays[0][1], lds[0], Gs[0]
        S0, M0 = S * G0, S * G0 * B
        Kc, Kc8 = N0 * Cc, rk(N0 * Cc + 1)       # + the ones column that turns dW = A^T . dE into [weights; bias]
        ws = dict(sid=sid, B=B, T=T, L=L, S=S, M=M, Mk=Mk, C=Cc, Kc=Kc, Kc8=Kc8, N0=N0, out0=out0, ld0=ld0, G0=G0, M0=M0)
        ws['X'] = _f32(B, T, Cc, device=dev)
        ws['Y'] = _i32(B, L, device=dev)
        ws['lens'], ws['lens_d'] = _i32(B, device=dev), _i32(B, device=dev)
        ws['A'] = _bf(M0, Kc8, device=dev)
        ws['A'][:, Kc] = 1.0                      # never written by e2t_conv_pack; the weight image has a zero there
        if len(lays) == 1:
            ws['AT'] = _bf(Kc + 1, Mk, device=dev)
            ws['AT'][Kc, :M] = 1.0
        ws['E'] = _bf(M, self.F8, device=dev)
        if self.F8 > s.enc_embed:
            ws['E'][:, s.enc_embed] = 1.0         # ones column for layer 0's dW_x (see _Lstm.bwd_weights)
        ws['dEpre'] = _bf(M, self.F8, device=dev)
        ws['dEpreT'] = _bf(s.enc_embed, Mk, device=dev)
        # outputs / pre-activation gradients / decimated lengths of the conv layers below the top one
        ws['cv'] = []
        cum = 1
        for j in range(len(lays) - 1):
            ci, co, n = lays[j]
            cum *= n
            Mj = S * Gs[j] * B
            Ej = _bf(Mj, lds[j], device=dev)
            Ej[:, r8(co)] = 1.0                   # ones column: A^T . dE of the layer above yields its bias gradient there
            ws['cv'].append(dict(E=Ej, dEpre=_bf(Mj, lds[j], device=dev), lens=_i32(B, device=dev), M=Mj, cumN=cum))
        ws['E0'] = ws['cv'][0]['E'] if ws['cv'] else ws['E']
        ws['dEpre0'] = ws['cv'][0]['dEpre'] if ws['cv'] else ws['dEpre']
        ws['enc'] = [lay.alloc(S, B) for lay in self.enc]
        ws['dY'] = [_f32(M, lay.ldy, device=dev) for lay in self.enc]
        if self.aux:
            cat = s.aux_dist == 'categorical'
            ws['auxT'] = _i32(B, T, device=dev) if cat else _f32(B, T, s.aux_dim, device=dev)
            ws['tlens'], ws['tlens_d'], ws['nval'] = _i32(B, device=dev), _i32(B, device=dev), _i32(1, device=dev)
            ws['At'] = _i32(M, device=dev) if cat else _f32(M, s.aux_dim, device=dev)
            ws['aux'] = self.aux.alloc(M)
            ws['dP'] = _bf(M, rk(s.aux_dim), device=dev)
            ws['aux_rowloss'] = _f32(M, device=dev)
        Md = L * B
        ws['Md'] = Md
        ws['dlens'], ws['ntok'] = _i32(B, device=dev), _i32(1, device=dev)
        ws['cnt_g'] = _i32(2 + len(self.aux_x), device=dev)     # data parallel: GLOBAL (all ranks) token / aux-sample counts of the batch
        ws['auxx'] = []
        for ax, hx in zip(self.aux_x, s.aux_extra):
            cat = hx.get('dist', 'Gaussian') == 'categorical'
            ws['auxx'].append(dict(
                T=(_i32(B, T, device=dev) if cat else _f32(B, T, hx['dim'], device=dev)), tlens=_i32(B, device=dev),
                tlens_d=_i32(B, device=dev), nval=_i32(1, device=dev),
                At=(_i32(M, device=dev) if cat else _f32(M, hx['dim'], device=dev)), ff=ax.alloc(M),
                dP=_bf(M, rk(hx['dim']), device=dev), rowloss=_f32(M, device=dev), loss=_f32(1, device=dev), cat=cat))
        ws['U'], ws['Tg'] = _i32(Md, device=dev), _i32(Md, device=dev)
        ws['e'] = _bf(Md, self.E8, device=dev)
        if self.E8 > s.dec_embed:
            ws['e'][:, s.dec_embed] = 1.0             # ones column for the decoder's dW_x
        ws['dec'] = self.dec.alloc(L, B)
        ws['c0'] = _f32(B, s.dec_rnn, device=dev)
        ws['proj'] = self.proj.alloc(Md)
        ws['dlogits'] = _bf(Md, rk(s.vocab), device=dev)
        ws['dHd'] = _f32(Md, self.dec.ldy, device=dev)
        ws['de'] = _f32(Md, self.E8, device=dev)
        ws['dh0'], ws['dc0'] = _f32(B, s.dec_rnn, device=dev), _f32(B, s.dec_rnn, device=dev)
        ws['rowloss'], ws['correct'] = _f32(Md, device=dev), _f32(Md, device=dev)
        ws['pred'] = _i32(Md, device=dev)
        ws['loss'] = _f32(4, device=dev)           # decoder CE, aux, accuracy, (unused)
        ws['done'], ws['hyp'] = _i32(B, device=dev), _i32(B, L, device=dev)
        ws['graph'] = {}
        self._ws[key] = ws
        return ws

    @staticmethod
    def check_end_padded(X):
        """Raise if an utterance of the padded batch X [n,T,C] has an all-zero row in front of a non-zero one: lengths
        are read off the zero padding (trainers.py:806-807) and the device searches it from the tail."""
        nz = np.abs(np.asarray(X)).max(axis=2) > 0
        lens = nz.shape[1] - np.argmax(nz[:, ::-1], axis=1)
        lens[~nz.any(axis=1)] = 0
        if int(nz.sum()) != int(lens.sum()):
            bad = np.nonzero(nz.sum(1) != lens)[0]
            raise ValueError('utterances %s contain all-zero sample rows in front of their last valid row; zero rows are '
                             'reserved for the end padding (subjects.py:386-390)' % bad[:8].tolist())

    def set_batch(self, ws, batch):
        self.check_end_padded(batch['encoder_inputs'])
        ws['packed'] = False
        ws['have'] = None
        ws['X'].copy_(torch.as_tensor(np.asarray(batch['encoder_inputs']), dtype=torch.float32))
        ws['Y'].copy_(torch.as_tensor(np.asarray(batch['decoder_targets']), dtype=torch.int32))
        if self.aux and 'encoder_targets' in batch:
            ws['auxT'].copy_(torch.as_tensor(np.asarray(batch['encoder_targets']), dtype=ws['auxT'].dtype))
        for wx, tg in zip(ws['auxx'], batch.get('encoder_targets_extra') or []):
            wx['T'].copy_(torch.as_tensor(np.asarray(tg), dtype=wx['T'].dtype))

    # ------------------------------------------------------------------ bf16-staged inputs ("bf16 in")
    def packed_inputs_ok(self, sid):
        """bf16-staged inputs exist for a single-layer front-end (the im2row rows of a conv stack are in grouped order)."""
        return len(self.conv[sid]) == 1

    def pack_inputs(self, sid, X):
        """Stage a partition once per fit: X (device, fp32 [n][T][C], end-padded: trainers.py:808-818, subjects.py:386-390) ->
        the bf16 im2row rows the front-end multiplies -- time-reversed over each utterance's valid prefix, N samples per row, the
        ones column at N*C, zero beyond the length: exactly what e2t_conv_pack writes per step, so every later result is
        bit-identical to the fp32-staged path -- as PA [S][n][rk(N*C + 1)] (block t' = decimated step t' of every utterance),
        plus the lengths.  A step then reads 2 B per input sample (forward product) + 2 B (conv weight gradient) instead of
        4 B + 2 B written + 2 B read; load_packed_batch() assembles a batch's operand with one blocked row gather."""
        assert self.packed_inputs_ok(sid), 'bf16-staged inputs: single-layer temporal convolution only'
        s, dev = self.spec, self.device
        assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous() and X.dim() == 3 and X.shape[2] == s.channels[sid]
        n, T, Cc = (int(v) for v in X.shape)
        N = s.decimation
        S, Kc8 = ceil_div(T, N), rk(N * Cc + 1)
        lens, lens_d = _i32(n, device=dev), _i32(n, device=dev)
        PA = torch.empty(S, n, Kc8, dtype=torch.bfloat16, device=dev)       # (every element is written: zero fill + ones column)
        st = self.stream
        lib.e2t_seq_lengths_tail_f32(X.data_ptr(), n, T, Cc, N, lens.data_ptr(), lens_d.data_ptr(), st)
        lib.e2t_conv_pack(X.data_ptr(), lens.data_ptr(), n, T, Cc, N, PA.data_ptr(), Kc8, st)
        return dict(PA=PA, lens=lens, lens_d=lens_d, n=n, S=S, T=T, Kc8=Kc8, sid=sid)

    def load_packed_batch(self, ws, pk, idx_dev):
        """Rows idx_dev (device int32 [B]; -1 = padding utterance) of a pack_inputs() partition into the workspace: the
        time-major conv operand ws['A'] (one blocked gather: block t' of PA -> rows t'*B .. of A) and both length vectors.
        The front-end of the following forward passes starts at the product (no lengths pass, no pack, no x)."""
        B, S = ws['B'], ws['S']
        assert pk['S'] == S and pk['Kc8'] == ws['Kc8'] and pk['sid'] == ws['sid'] and ws['G0'] == 1
        rw = ws['Kc8'] // 2                         # 32-bit words per row
        st = self.stream
        lib.e2t_gather_rows_blocks_u32(pk['PA'].data_ptr(), idx_dev.data_ptr(), B, B, rw, S, pk['n'] * rw, B * rw, ws['A'].data_ptr(), st)
        lib.e2t_gather_rows_u32(pk['lens'].data_ptr(), idx_dev.data_ptr(), B, B, 1, ws['lens'].data_ptr(), st)
        lib.e2t_gather_rows_u32(pk['lens_d'].data_ptr(), idx_dev.data_ptr(), B, B, 1, ws['lens_d'].data_ptr(), st)
        ws['packed'] = True
        ws['A_stale'] = False

    def set_prefetch(self, ws, sources):
        """Round 6: let the CAPTURED train step of this workspace assemble the NEXT batch itself.  sources = [(resident partition
        tensor [n, ...], workspace buffer [B, ...])] (the fit's HBM-resident arrays: inputs, decoder targets, encoder targets); the
        rows to take are ws['next_idx'] (device int32 [B], -1 = padding utterance), which the caller fills before every replay.
        The gathers sit on the side branch behind the decoder-side preparation (the only reader of the target buffers) and behind the
        front-end (the only reader of the inputs): a batch's 105 MB (cfg2) move under the encoder instead of between two steps.
        None switches it off.  Only captured steps prefetch (ws['prefetched'] says whether the last train_step did)."""
        if sources is None or not self.options['prefetch_batches'] or not self.overlap or self.aux_x:
            ws.pop('prefetch', None)
            return False
        if 'next_idx' not in ws:
            ws['next_idx'] = torch.full((ws['B'],), -1, dtype=torch.int32, device=self.device)
        ws['prefetch'] = [(src, dst, int(src[0].numel())) for src, dst in sources]
        ws['prefetch_key'] = tuple((src.data_ptr(), dst.data_ptr(), int(src.shape[0])) for src, dst in sources)
        return True

    def _prefetch_next(self, ws):
        st = self.stream
        for src, dst, words in ws['prefetch']:
            lib.e2t_gather_rows_u32(src.data_ptr(), ws['next_idx'].data_ptr(), ws['B'], ws['B'], words, dst.data_ptr(), st)

    def set_global_counts(self, ws, ntok, nval=0, nval_extra=()):
        """Data parallel: the batch's token count and auxiliary-sample count(s) over ALL ranks (host integers; every rank
        holds the targets of the whole global batch, or sums its own counts over the ranks once)."""
        extra = [max(int(v), 1) for v in nval_extra] + [1] * (len(self.aux_x) - len(nval_extra))
        ws['cnt_g'].copy_(torch.tensor([max(int(ntok), 1), max(int(nval), 1)] + extra, dtype=torch.int32))
        ws['global_counts'] = True

    def local_counts(self, batch_Y, batch_A=None, extra=()):
        """(tokens, auxiliary samples) this rank's slice contributes, counted on the host exactly as the kernels do:
        non-pad target tokens; ceil(valid target length / decimation) per utterance (non-zero rows / non-pad ids)."""
        Y = np.asarray(batch_Y)
        ntok = int((Y != PAD_ID).sum())
        nval = 0
        if batch_A is not None and self.aux is not None:
            A = np.asarray(batch_A)
            tl = (A != PAD_ID).sum(1) if A.ndim == 2 else (np.abs(A).max(axis=2) > 0).sum(1)
            nval = int((-(-tl // self.spec.decimation)).sum())
        if extra:
            nx = []
            for A in extra:
                A = np.asarray(A)
                tl = (A != PAD_ID).sum(1) if A.ndim == 2 else (np.abs(A).max(axis=2) > 0).sum(1)
                nx.append(int((-(-tl // self.spec.decimation)).sum()))
            return ntok, nval, nx
        return ntok, nval

    # ------------------------------------------------------------------ forward
    def encode(self, ws, src, train, after_layer=None, after_first=None, after_gx=None, before_weights=None, before_enc=None):
        s = self.spec
        B, T, S, M, Cc, N = ws['B'], ws['T'], ws['S'], ws['M'], ws['C'], s.decimation
        st = self.stream
        packed = bool(ws.get('packed'))       # bf16-staged inputs (load_packed_batch): ws['A'] and the lengths are the batch
        # (searched from the tail: end-padded batches only -- subjects.py:386-390; set_batch / the staging code check that)
        if not packed:
            lib.e2t_seq_lengths_tail_f32(ws['X'].data_ptr(), B, T, Cc, N, ws['lens'].data_ptr(), ws['lens_d'].data_ptr(), st)
        if after_first is not None:
            after_first()
        stack = len(self.conv[ws['sid']]) > 1
        assert not (stack and packed)
        fused = (self.fused_conv == '1' or (self.fused_conv == 'auto' and B * T * Cc * 4 >= (1 << 28))) \
            and bool(H.load().e2t_conv_fwd_fused_ok(Cc, s.enc_embed)) and not stack and not packed
        if stack:
            self._conv_stack_fwd(ws, src, train, before_weights)
        elif not fused and not packed:
            lib.e2t_conv_pack(ws['X'].data_ptr(), ws['lens'].data_ptr(), B, T, Cc, N, ws['A'].data_ptr(), ws['Kc8'], st)
        ws['A_stale'] = fused and not train   # (inference through the fused kernel leaves no im2row copy; a backward pass after it packs first)
        if before_weights is not None and not stack:
            before_weights()
        if stack:
            pass
        elif fused:
            # one pass over the fp32 electrode grid: reversal + im2row + bf16 rounding in the GEMM's staging path
            ep = H.GemmEpilogue()
            ep.bias = self.store.ptr('conv%s.W' % ws['sid'], src, ws['Kc'] * s.enc_embed)
            ep.alpha = 1.0
            ep.flags = (H.GEMM_RELU if s.conv_relu else 0) | H.GEMM_OUT_BF16
            ep.row_lens, ep.rows_per_step = ws['lens_d'].data_ptr(), B
            if train and s.ff_dropout > 0:
                ep.flags |= H.GEMM_DROPOUT
                ep.drop_rate, ep.drop_stream, ep.drop_ld = s.ff_dropout, STREAM_CONV, s.enc_embed
                ep.drop_seed, ep.drop_step = self.seed, self.step_t.data_ptr()
            ep.splitk_ws, ep.splitk_ws_bytes = self.splitk_ws.data_ptr(), self.splitk_ws.numel() * 4
            # training: the packed im2row copy the conv weight gradient reads is emitted on the way (no second pass over x)
            lib.e2t_conv_fwd_fused(ws['X'].data_ptr(), ws['lens'].data_ptr(), B, T, Cc, N, self.convT[ws['sid']].data_ptr(),
                                   ws['Kc8'], ws['E'].data_ptr(), self.F8, s.enc_embed,
                                   ws['A'].data_ptr() if train else None, ws['Kc8'], C.byref(ep), st)
        else:
            self.gemm(ws['A'].data_ptr(), ws['Kc8'], self.convT[ws['sid']].data_ptr(), ws['Kc8'], ws['E'].data_ptr(), self.F8,
                      M, s.enc_embed, ws['Kc8'],
                      bias=self.store.ptr('conv%s.W' % ws['sid'], src, ws['Kc'] * s.enc_embed), relu=s.conv_relu, out_bf16=True,
                      drop=(s.ff_dropout if train else 0.0, STREAM_CONV, s.enc_embed), row_lens=(ws['lens_d'].data_ptr(), B),
                      alg=(M, s.enc_embed, ws['Kc']))
        x = ws['E'].data_ptr()
        if before_enc is not None:
            before_enc()
        for l, (lay, lw) in enumerate(zip(self.enc, ws['enc'])):
            lay.fwd(lw, x, ws['lens_d'], src, train, after_gx=(lambda l=l: after_gx(l)) if after_gx is not None else None)
            x = lw['Ydrop'].data_ptr()
            if after_layer is not None:
                after_layer(l)
        last, lw = self.enc[-1], ws['enc'][-1]
        # encoder final state -> block 0 of the decoder's ext output array, and c0
        lib.e2t_final_state(lw['Yext'].data_ptr(), last.ldy, lw['Cs'].data_ptr(), ws['lens_d'].data_ptr(), B, last.H,
                            ws['dec']['Yext'].data_ptr(), self.dec.ldy, ws['c0'].data_ptr(), st)

    # ------------------------------------------------------------------ conv stack (spec.conv_pre)
    def _conv_stack_fwd(self, ws, src, train, before_weights=None):
        """Several strided conv layers (trainers.py:406-407: their strides multiply to the decimation factor; width == stride,
        :535-541).  Layer 0 is the im2row product over x with its rows in grouped order; layer j >= 1 multiplies a VIEW of the
        layer below ([M_j][N_j * ld_{j-1}]) with an operand image that has the weights of tap w at columns w * ld_{j-1}."""
        s, sid, st = self.spec, ws['sid'], self.stream
        B, T, Cc = ws['B'], ws['T'], ws['C']
        lays, lds, Gs = self.conv[sid], self.conv_ld[sid], self.conv_G[sid]
        for cv in ws['cv']:          # decimated lengths after every layer below the top one (the top one's are lens_d)
            lib.e2t_seq_lengths_tail_f32(ws['X'].data_ptr(), B, T, Cc, cv['cumN'], ws['lens'].data_ptr(), cv['lens'].data_ptr(), st)
        lib.e2t_conv_pack_grouped(ws['X'].data_ptr(), ws['lens'].data_ptr(), B, T, Cc, ws['N0'], ws['G0'], ws['A'].data_ptr(), ws['Kc8'], st)
        if before_weights is not None:
            before_weights()
        rate = s.ff_dropout if train else 0.0
        top = len(lays) - 1
        for j, (ci, co, n) in enumerate(lays):
            out = ws['E'] if j == top else ws['cv'][j]['E']
            lens = ws['lens_d'] if j == top else ws['cv'][j]['lens']
            Mj = ws['M'] if j == top else ws['cv'][j]['M']
            if j == 0:
                a_ptr, lda, b_ptr, K, kalg = ws['A'].data_ptr(), ws['Kc8'], self.convT[sid].data_ptr(), ws['Kc8'], ws['Kc']
            else:
                K = n * lds[j - 1]
                a_ptr, lda, b_ptr, kalg = ws['cv'][j - 1]['E'].data_ptr(), K, self.convTj[sid][j].data_ptr(), n * ci
            self.gemm(a_ptr, lda, b_ptr, K, out.data_ptr(), lds[j], Mj, co, K,
                      bias=self.store.ptr(conv_seg(sid, j), src, n * ci * co), relu=True, out_bf16=True,
                      drop=(rate, STREAM_CONV if j == top else STREAM_CONV_PRE + j, co), row_lens=(lens.data_ptr(), B), row_group=Gs[j],
                      alg=(Mj, co, kalg))

    def _conv_stack_bwd(self, ws):
        """Weight gradients of the conv layers above the bottom one and the pre-activation gradient handed down to each
        layer below (ends with ws['dEpre0'], which the bottom layer's dK = A^T . dEpre and the saliency path read)."""
        s, sid, store = self.spec, ws['sid'], self.store
        lays, lds = self.conv[sid], self.conv_ld[sid]
        train = ws.get('fwd_train', True)
        keep = 1.0 / (1.0 - s.ff_dropout) if (train and s.ff_dropout > 0) else 1.0
        top = len(lays) - 1
        for j in range(top, 0, -1):
            ci, co, n = lays[j]
            dEp = ws['dEpre'] if j == top else ws['cv'][j]['dEpre']
            Mj = ws['M'] if j == top else ws['cv'][j]['M']
            prev, ldp = ws['cv'][j - 1], lds[j - 1]
            g = store.ptr(conv_seg(sid, j), store.g)
            # dW of tap w = (columns w*ldp .. of the view)^T . dEpre: one batched K-major product, straight into the segment
            self.gemm(prev['E'].data_ptr(), n * ldp, dEp.data_ptr(), lds[j], g, co, ci, co, Mj, splitk=True, tn=True,
                      batch=(n, ldp, 0, ci * co), alg=(ci, co, Mj))
            # bias gradient: the ones column of the layer below (index r8(in), 16-B aligned) against dEpre
            self.gemm(prev['E'].data_ptr() + 2 * r8(ci), n * ldp, dEp.data_ptr(), lds[j], g + 4 * n * ci * co, co, 1, co, Mj,
                      splitk=True, tn=True, alg=(1, co, Mj))
            # d(pre-activation) of the layer below, through the view: [M_j][n*ldp] = dEpre . W (masked by E != 0, x 1/keep)
            self.gemm(dEp.data_ptr(), lds[j], self.convBj[sid][j].data_ptr(), lds[j], prev['dEpre'].data_ptr(), n * ldp, Mj, n * ldp, lds[j],
                      out_bf16=True, alpha=keep, mask_src=(prev['E'].data_ptr(), n * ldp), alg=(Mj, n * ci, co))

    def forward(self, ws, train=True, which=None, with_aux=True, pack_first=False, global_counts=False, pack_skip=None):
        """Teacher-forced forward incl. losses and d(logits); leaves everything backward needs in ws.
        global_counts: normalise the losses by the counts in ws['cnt_g'] (set_global_counts: the token / auxiliary-sample
        counts of the batch over ALL ranks) instead of this rank's own, so that the SUM of the ranks' gradients is the
        gradient of the global mean loss whatever the shard sizes are."""
        s = self.spec
        src = getattr(self.store, which or 'p')
        B, T, L, S, M, Md, N = ws['B'], ws['T'], ws['L'], ws['S'], ws['M'], ws['Md'], s.decimation
        st = self.stream
        ws['use_aux'] = bool(self.aux and with_aux and s.aux_scale != 0.0)
        cat = s.aux_dist == 'categorical'
        ntokp = ws['cnt_g'].data_ptr() if global_counts else ws['ntok'].data_ptr()
        nvalp = (ws['cnt_g'].data_ptr() + 4) if global_counts else (ws['nval'].data_ptr() if self.aux else None)

        def aux_targets():
            st = self.stream
            if cat:
                lib.e2t_seq_lengths_i32(ws['auxT'].data_ptr(), B, T, PAD_ID, N, ws['tlens'].data_ptr(), ws['tlens_d'].data_ptr(), st)
                lib.e2t_gather_rev_decim_i32(ws['auxT'].data_ptr(), ws['tlens'].data_ptr(), B, T, N, ws['At'].data_ptr(), st)
            else:
                lib.e2t_seq_lengths_f32(ws['auxT'].data_ptr(), B, T, s.aux_dim, N, ws['tlens'].data_ptr(), ws['tlens_d'].data_ptr(), st)
                lib.e2t_gather_rev_decim_f32(ws['auxT'].data_ptr(), ws['tlens'].data_ptr(), B, T, s.aux_dim, N, ws['At'].data_ptr(), st)
            lib.e2t_sum_i32(ws['tlens_d'].data_ptr(), B, ws['nval'].data_ptr(), st)

        def aux_forward():
            st = self.stream
            k = s.aux_layer
            out = self.aux.fwd(ws['aux'], ws['enc'][k]['Ydrop'].data_ptr(), src, train)
            if cat:
                lib.e2t_softmax_ce(out.data_ptr(), s.aux_dim, M, s.aux_dim, ws['At'].data_ptr(), ws['tlens_d'].data_ptr(), B,
                                   nvalp, s.aux_scale, ws['aux_rowloss'].data_ptr(), None, None,
                                   ws['dP'].data_ptr(), rk(s.aux_dim), st)
                lib.e2t_sum_f32(ws['aux_rowloss'].data_ptr(), M, nvalp, 1.0, ws['loss'].data_ptr() + 4, st)
            else:
                lib.e2t_mse(out.data_ptr(), s.aux_dim, ws['At'].data_ptr(), M, s.aux_dim, ws['tlens_d'].data_ptr(), B,
                            nvalp, s.aux_scale, ws['aux_rowloss'].data_ptr(), ws['dP'].data_ptr(),
                            rk(s.aux_dim), st)
                lib.e2t_sum_f32(ws['aux_rowloss'].data_ptr(), M, nvalp, 1.0 / s.aux_dim,
                                ws['loss'].data_ptr() + 4, st)

        def dec_prep():
            # teacher-forced decoder inputs (tokens, embedding, input projection) and the auxiliary targets: nothing of
            # the encoder in them, so they run on the side stream next to the (HBM-bound) front-end instead of between
            # encoder and decoder
            st = self.stream
            lib.e2t_seq_lengths_i32(ws['Y'].data_ptr(), B, L, PAD_ID, 1, ws['dlens'].data_ptr(), None, st)
            lib.e2t_sum_i32(ws['dlens'].data_ptr(), B, ws['ntok'].data_ptr(), st)
            lib.e2t_decoder_tokens(ws['Y'].data_ptr(), B, L, EOS_ID, ws['U'].data_ptr(), ws['Tg'].data_ptr(), st)
            dr = self._dropout(s.ff_dropout if train else 0.0, STREAM_DEC_EMB)
            lib.e2t_embed_fwd(self.emb.data_ptr(), self.E8, ws['U'].data_ptr(), 0, Md, s.dec_embed, ws['e'].data_ptr(), self.E8,
                              C.byref(dr), st)
            self.dec.fwd_gx(ws['dec'], ws['e'].data_ptr(), src)
            if ws['use_aux']:
                aux_targets()
        ahead = self.overlap
        joins = []
        pend = {}
        ev0 = self.fork_point() if ahead else None

        def after_first():
            # (side work is enqueued AFTER the main branch's next kernel: see fork_point)
            if ahead:
                if pack_first:
                    # the operand re-pack of the optimiser step that came before runs here, next to the weight-free
                    # start of the front-end (lengths, im2row) instead of in front of it
                    def pack_side():
                        self.pack(which or 'p', after_head=lambda: pend.__setitem__('pack_head', self.fork_point()), skip_ranges=pack_skip)
                    pend['pack'] = self.run_side(ev0, pack_side)
                pend['dec'] = self.run_side(ev0, dec_prep)

        def before_weights():
            # the conv GEMM needs the conv images only (first pack launch); everything else is joined in front of the
            # first encoder layer's input projection
            if 'pack_head' in pend:
                self.join_side(pend.pop('pack_head'))

        def before_enc():
            if 'pack' in pend:
                self.join_side(pend.pop('pack'))
            if len(self.enc) < 2:
                prefetch_mark()

        def prefetch_mark():
            # (two phases, as everywhere in a captured step: the fork point is taken HERE, the side work is enqueued behind the main
            #  branch's next kernel -- of the two children of a fork the one created first keeps the parent's hardware queue)
            if ahead and train and ws.get('prefetch') and ws.get('_prefetch_now'):
                pend['pf_ev'] = self.fork_point()

        def prefetch_here():
            if 'pf_ev' in pend:
                # every reader of this batch's inputs (lengths, im2row / the one-pass front-end) is enqueued on this branch, the
                # readers of its targets (dec_prep) on the side stream: the next batch may land (set_prefetch).  WHERE: behind
                # the bottom layer's recurrence, beside the next layer's input projection.  MEASURED AND LEFT OFF (scripts/bench_fit.py,
                # three same-box pairs each): forked right behind the front-end 1.775 vs 1.69 ms per step; here, created before the main
                # branch's next kernel 1.85 vs 1.70; here in two phases 1.78 vs 1.68 -- 210 MB of copy traffic beside a persistent
                # recurrence sit in the memory queues of the CUs that hand the state around (MI355X_MICROARCH.md, handoff-1to1 by
                # streaming waves on the endpoint CUs) and cost 140 us where the gather between two steps costs 42.
                joins.append(self.run_side(pend.pop('pf_ev'), lambda: self._prefetch_next(ws)))

        def after_gx(l):
            prefetch_here()
            # the auxiliary head taps layer aux_layer: its forward starts once the NEXT layer's input projection is done,
            # i.e. under that layer's recurrence (latency-bound, 56 CUs idle) rather than next to the projection GEMM
            # (which it slowed from 69 to 96 us), and long before the decoder, which then has the chip to itself
            if ahead and ws['use_aux'] and l == s.aux_layer + 1:
                pend['aux_ev'] = self.fork_point()

        def after_layer(l):
            if l == 0 and len(self.enc) >= 2:
                prefetch_mark()
            if ahead and ws['use_aux'] and l == s.aux_layer and l == len(self.enc) - 1:
                pend['aux_ev'] = self.fork_point()         # the head taps the top layer: under the decoder
            elif 'aux_ev' in pend:
                joins.append(self.run_side(pend.pop('aux_ev'), aux_forward))
        if (pack_first and not ahead) or (not pack_first and self._packed != (which or 'p')):
            self.pack(which or 'p')        # (a captured train step leaves the images one update behind: see train_step)
        self.encode(ws, src, train, after_layer, after_first, after_gx, before_weights, before_enc)
        if 'aux_ev' in pend:
            joins.append(self.run_side(pend.pop('aux_ev'), aux_forward))
        for j, (ax, hx, wx) in enumerate(zip(self.aux_x, s.aux_extra, ws['auxx'])):
            wx['use'] = bool(with_aux and hx.get('scale', 1.0) != 0.0)
            if not wx['use']:
                continue
            # further auxiliary heads: targets (lengths, reversal, decimation), stack, loss + d(output) -- on the main stream
            if wx['cat']:
                lib.e2t_seq_lengths_i32(wx['T'].data_ptr(), B, T, PAD_ID, N, wx['tlens'].data_ptr(), wx['tlens_d'].data_ptr(), st)
                lib.e2t_gather_rev_decim_i32(wx['T'].data_ptr(), wx['tlens'].data_ptr(), B, T, N, wx['At'].data_ptr(), st)
            else:
                lib.e2t_seq_lengths_f32(wx['T'].data_ptr(), B, T, hx['dim'], N, wx['tlens'].data_ptr(), wx['tlens_d'].data_ptr(), st)
                lib.e2t_gather_rev_decim_f32(wx['T'].data_ptr(), wx['tlens'].data_ptr(), B, T, hx['dim'], N, wx['At'].data_ptr(), st)
            lib.e2t_sum_i32(wx['tlens_d'].data_ptr(), B, wx['nval'].data_ptr(), st)
            nvp = (ws['cnt_g'].data_ptr() + 4 * (2 + j)) if global_counts else wx['nval'].data_ptr()
            out = ax.fwd(wx['ff'], ws['enc'][hx['layer']]['Ydrop'].data_ptr(), src, train)
            sc = float(hx.get('scale', 1.0))
            if wx['cat']:
                lib.e2t_softmax_ce(out.data_ptr(), hx['dim'], M, hx['dim'], wx['At'].data_ptr(), wx['tlens_d'].data_ptr(), B, nvp, sc,
                                   wx['rowloss'].data_ptr(), None, None, wx['dP'].data_ptr(), rk(hx['dim']), st)
                lib.e2t_sum_f32(wx['rowloss'].data_ptr(), M, nvp, 1.0, wx['loss'].data_ptr(), st)
            else:
                lib.e2t_mse(out.data_ptr(), hx['dim'], wx['At'].data_ptr(), M, hx['dim'], wx['tlens_d'].data_ptr(), B, nvp, sc,
                            wx['rowloss'].data_ptr(), wx['dP'].data_ptr(), rk(hx['dim']), st)
                lib.e2t_sum_f32(wx['rowloss'].data_ptr(), M, nvp, 1.0 / hx['dim'], wx['loss'].data_ptr(), st)
        jdec = pend.get('dec')
        if ws['use_aux'] and not ahead:
            aux_targets()
            aux_forward()
        # decoder (teacher forced)
        if jdec is not None:
            self.join_side(jdec)
        else:
            dec_prep()
        self.dec.fwd(ws['dec'], ws['e'].data_ptr(), ws['dlens'], src, train, c0=ws['c0'], gx_done=True)
        logits = self.proj.fwd(ws['proj'], ws['dec']['Ydrop'].data_ptr(), src, train)
        lib.e2t_softmax_ce(logits.data_ptr(), s.vocab, Md, s.vocab, ws['Tg'].data_ptr(), ws['dlens'].data_ptr(), B,
                           ntokp, s.dec_scale, ws['rowloss'].data_ptr(), ws['pred'].data_ptr(),
                           ws['correct'].data_ptr(), ws['dlogits'].data_ptr(), rk(s.vocab), st)
        # (decoder loss and token accuracy in ONE launch: two single-workgroup launches sat on the critical branch here)
        if self.options['lean_critical']:
            lib.e2t_sum2_f32(ws['rowloss'].data_ptr(), ws['correct'].data_ptr(), Md, ntokp, 1.0, 1.0, ws['loss'].data_ptr(), ws['loss'].data_ptr() + 8, st)
        else:
            lib.e2t_sum_f32(ws['rowloss'].data_ptr(), Md, ntokp, 1.0, ws['loss'].data_ptr(), st)
            lib.e2t_sum_f32(ws['correct'].data_ptr(), Md, ntokp, 1.0, ws['loss'].data_ptr() + 8, st)
        for j in joins:
            self.join_side(j)

    # ------------------------------------------------------------------ backward
    def backward_stages(self, ws):
        """[(main, side, [(a,b) grad ranges finished by the stage])].  The critical path of the backward pass is
        head -> BPTT(top) -> dX -> BPTT(next) -> ...; the weight gradients of a layer (operand transposes + split-K
        GEMMs, ~as long as a BPTT sweep) depend only on that layer's dG, so stage k runs BPTT + input gradient of
        layer l on the main stream and the weight gradients of layer l+1 on a side stream (a parallel branch of the
        captured hipGraph).  The persistent recurrence is latency-bound and leaves most MFMA cycles (and 32-56 CUs)
        idle; the side stream's GEMMs run on those CUs (the K-major instance needs more registers than a CU with a
        BPTT workgroup has left: DESIGN.md 8.4)."""
        store = self.store
        nl = len(self.enc)
        stages = []

        def rng_of(names):
            out = []
            for nm in names:
                a, b = store.seg_range(nm)
                if out and out[-1][1] == a:
                    out[-1][1] = b
                else:
                    out.append([a, b])
            return [tuple(r) for r in out]
        head = [n for n in store.order if n.startswith('proj') or n.startswith('dec.')]
        enc_names = lambda l: [n for n in store.order if n.startswith('enc%d.' % l)]
        aux_names = [n for n in store.order if n.startswith('aux')]
        conv_names = [conv_seg(ws['sid'], j) for j in range(len(self.conv[ws['sid']]) - 1, -1, -1)]
        # the auxiliary head's backward (its own weight gradients + its share of dY[aux_layer]) only needs the forward
        # pass: side stream, under the decoder's BPTT; the layer above then ACCUMULATES its input gradient onto it
        stages.append((lambda train: self._bwd_head(ws, train), lambda train: self._bwd_aux(ws, train), rng_of(aux_names)))
        for l in range(nl - 1, -1, -1):
            if l < nl - 1:
                names = enc_names(l + 1)
                side = (lambda train, l=l: self._bwd_enc_weights(ws, l + 1))
            else:       # the head's own weight gradients queue up first, under the top layer's BPTT
                names = head
                side = (lambda train: self._bwd_head_weights(ws, train))
            stages.append((lambda train, l=l: self._bwd_enc_rec(ws, l, train), side, rng_of(names)))
        stages.append((lambda train: self._bwd_enc_weights(ws, 0), None, rng_of(enc_names(0) + conv_names)))
        return stages

    def fork_side(self, fn):
        """Run fn() on the side stream, ordered after everything enqueued so far on the current stream (a parallel
        branch of a captured hipGraph).  Returns the event to pass to join_side()."""
        cur = torch.cuda.current_stream(self.device)
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(device=self.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        self._wstream.wait_event(fork)
        with torch.cuda.stream(self._wstream):
            self._on_side = True
            try:
                fn()
            finally:
                self._on_side = False
            join = torch.cuda.Event()
            join.record(self._wstream)
        return join

    def fork_point(self):
        """Event marking 'everything enqueued so far on the current stream'; run_side(ev, fn) later hangs fn off it.  Two
        phases because the ORDER of enqueueing matters inside a captured graph: the branch whose first node is created
        first keeps the parent's hardware queue, the other one pays a cross-queue hand-off (~10 us, measured) -- so the
        critical branch is enqueued first and the side work afterwards."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def run_side(self, ev, fn, stream=None, also=()):
        """fn() on the side stream (or `stream`), ordered after fork_point() event ev.  Returns the event to pass to
        join_side().  (A fork off a forked stream crashes hipGraphInstantiate: side work always forks from the MAIN branch.)"""
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(device=self.device)
        stream = stream or self._wstream
        stream.wait_event(ev)
        for e in also:                    # further events the work waits for (of other side branches)
            stream.wait_event(e)
        with torch.cuda.stream(stream):
            self._on_side = True
            try:
                fn()
            finally:
                self._on_side = False
            join = torch.cuda.Event()
            join.record(stream)
        return join

    def join_side(self, join):
        torch.cuda.current_stream(self.device).wait_event(join)

    def run_stage(self, main, side, train):
        """main on the current stream, side (if any) on the side stream, joined at the end."""
        if side is None or not self.overlap:
            if side is not None:
                side(train)
            main(train)
            return
        join = self.fork_side(lambda: side(train))
        main(train)
        self.join_side(join)

    def backward(self, ws, train=True, after_stage=None, early=None, before_join=None, exchange=None, after_last_rec=None):
        """early = (stage index, fn): fn() is queued on the side stream behind that stage's side work AND behind the stage's
        main-branch work (the optimiser update of the parameter ranges whose gradients are complete by then: HBM-bound, next
        to the compute-bound weight-gradient GEMMs of the remaining stages; the stage is the last one with a recurrence, so
        every kernel that can raise `sync_err` has finished before the first update reads it).  before_join(): queued on the
        main stream behind its last stage, in front of the joins with the side stream (the update of the bottom layer, whose
        gradients the main chain produced itself, runs while the side stream is still busy with the early update and its
        re-pack).  exchange(ranges): data parallel inside ONE captured graph -- called on the stream that completes a stage's
        gradient ranges, right behind the work that completes them (the collective orders itself behind that stream).
        after_last_rec(): on the main stream behind the last recurrence of the backward pass."""
        ws['have_dy'] = [False] * len(self.enc)
        ws['_aux_join'] = None
        ws['fwd_train'] = train
        self._arena_seq = 0                 # (slab arenas are dealt in launch order: the same order in every backward pass)
        deferred = []
        pending_early = None
        stages = self.backward_stages(ws)
        last_rec = len(stages) - 2                  # stages: head + aux | one per encoder layer (top .. bottom) | bottom weights
        for i, (main, side, ranges) in enumerate(stages):
            def side_work(side=side, ranges=ranges):
                side(train)
                if exchange is not None:
                    exchange(ranges)
            if after_stage is None and self.overlap and side is not None and i == 0:
                # auxiliary head: joined where the main branch first touches dY[aux_layer] (_bwd_enc_rec)
                ev = self.fork_point()
                main(train)
                ws['_aux_join'] = self.run_side(ev, side_work)
            elif after_stage is None and self.overlap and side is not None and i > 0:
                # nobody needs a layer's weight gradients before the optimiser: the side stream just queues them (it is
                # ~1.4x longer than the BPTT chain) and is joined once at the end instead of after every stage
                ev = self.fork_point()
                main(train)
                deferred.append(self.run_side(ev, side_work))
            else:
                self.run_stage(main, side, train)
                if exchange is not None:
                    exchange(ranges)
            if i == last_rec and after_last_rec is not None:
                after_last_rec()
            if early is not None and i == early[0] and after_stage is None and self.overlap:
                # the update's fork point: behind this stage's recurrence (the last kernel that can raise sync_err) ...
                pending_early = (self.fork_point(), deferred[-1])
            if pending_early is not None and (i > early[0] or exchange is not None):
                # ... but its branch is CREATED only now, after the main branch's next nodes (fork_point / run_side: of the two
                # children of a fork the one created first keeps the parent's hardware queue; created first, the update took it
                # and the executor queued the whole weight-gradient branch behind the main chain: 1.62 -> 1.92 ms at cfg2).
                # (Data parallel, where the communicator's branch forks off the same point: measured the other way round --
                #  created at once 1.62 ms, created late 1.94 -- so there it is created at once.  Both are properties of how
                #  hipGraphLaunch deals branches to hardware queues, not of the step.)
                # Data parallel: the update waits for the communicator's stream, which waited for the side stream (weight
                # gradients -> all-reduce).  Two captured streams that wait for EACH OTHER send hipStreamEndCapture of ROCm 7.0
                # into an endless recursion (stack overflow inside libamdhip64: it walks the streams' wait relation, not the
                # node graph), so the update has a stream of its own, forked from the main branch: update -> communicator ->
                # side -> main is a chain.  It is ordered behind the main branch's fork point first (a stream must enter the
                # capture through the main branch: run_side), then behind the side stream's weight gradients.
                if self._ustream is None:
                    self._ustream = torch.cuda.Stream(device=self.device)
                deferred.append(self.run_side(pending_early[0], early[1], stream=self._ustream, also=(pending_early[1],)))
                pending_early = None
            if after_stage:
                after_stage(i, ranges)
        if ws.get('_aux_join') is not None:
            self.join_side(ws['_aux_join'])
            ws['_aux_join'] = None
        if before_join is not None:
            before_join()
        for j in deferred:
            self.join_side(j)

    def _bwd_head(self, ws, train):
        """Critical path of the head: gradient through the vocabulary projection, decoder BPTT (-> gradient into the
        encoder's final state and into the embedded tokens)."""
        s, store = self.spec, self.store
        if not self.options['lean_critical']:
            a_, b_ = store.seg_range('dec.emb')
            lib.e2t_fill_u32(store.g.data_ptr() + 4 * a_, b_ - a_, 0, self.stream)
        dd = self.dec.out_drop(train)
        self.proj.bwd_dx(ws['proj'], ws['dlogits'], ws['dHd'].data_ptr(), self.dec.ldy, False, train, d_in_drop=dd)
        self.dec.bwd_rec(ws['dec'], ws['e'].data_ptr(), ws['dlens'], ws['dHd'].data_ptr(), self.dec.ldy, train,
                         None, self.E8, c0=ws['c0'], dh0=ws['dh0'], dc0=ws['dc0'], dy_masked=dd is not None)

    def _bwd_head_weights(self, ws, train):
        """Weight gradients of the head (projection, decoder, embedding): nothing downstream needs them."""
        s, store = self.spec, self.store
        # projection, decoder input kernel and decoder recurrent kernel: one grouped launch (K = L*B rows each)
        with self.gemm_group():
            self.proj.bwd_dw(ws['proj'], ws['dec']['Ydrop'].data_ptr(), ws['dlogits'])
            self.dec.bwd_weights(ws['dec'], ws['e'].data_ptr())
        # the gradient into the embedded tokens only feeds the embedding table: off the encoder's critical path
        self.dec.bwd_d_in(ws['dec'], ws['de'].data_ptr(), self.E8)
        # (the embedding scatter-add accumulates by atomics onto a zeroed segment: zeroed HERE, on the branch that uses it -- it used
        #  to be the first launch of the backward pass on the critical branch)
        if self.options['lean_critical']:
            a_, b_ = store.seg_range('dec.emb')
            lib.e2t_fill_u32(store.g.data_ptr() + 4 * a_, b_ - a_, 0, self.stream)
        dr = self._dropout(s.ff_dropout if train else 0.0, STREAM_DEC_EMB)
        lib.e2t_embed_bwd(ws['de'].data_ptr(), self.E8, ws['U'].data_ptr(), ws['Md'], s.dec_embed,
                          store.ptr('dec.emb', store.g), s.dec_embed, C.byref(dr), self.stream)

    def _bwd_enc_rec(self, ws, l, train):
        """aux head (if it taps layer l), BPTT of layer l, gradient into the layer below."""
        s = self.spec
        nl = len(self.enc)
        have_dy = ws['have_dy']
        lay, lw = self.enc[l], ws['enc'][l]
        x = ws['E'].data_ptr() if l == 0 else ws['enc'][l - 1]['Ydrop'].data_ptr()
        dY = ws['dY'][l].data_ptr() if have_dy[l] else None
        fin = dict(dh_final=ws['dh0'], dc_final=ws['dc0']) if l == nl - 1 else {}
        aj = ws.get('_aux_join')
        if aj is not None and self.aux_x:
            # further heads tap other layers: their share of dY must be complete before ANY layer's BPTT or input gradient
            # touches it -- joined once, in front of the top layer
            ws['_aux_join'] = None
            self.join_side(aj)
        elif aj is not None and s.aux_layer is not None and (l == s.aux_layer or l == s.aux_layer + 1):
            # the auxiliary head's backward runs on the side stream since the start of the backward pass; it writes
            # dY[aux_layer], which this layer's input gradient accumulates onto (l = aux_layer + 1) or whose BPTT reads
            # (l = aux_layer, when the head taps the top layer)
            ws['_aux_join'] = None
            if l == s.aux_layer:
                self.join_side(aj)
            else:
                fin['before_d_in'] = lambda: self.join_side(aj)
        fin['dy_masked'] = lay.out_drop(train) is not None
        if l > 0:
            lay.bwd_rec(lw, x, ws['lens_d'], dY, lay.ldy, train, ws['dY'][l - 1].data_ptr(), self.enc[l - 1].ldy,
                        d_in_accumulate=have_dy[l - 1], d_in_drop=self.enc[l - 1].out_drop(train), **fin)
            have_dy[l - 1] = True
            return
        keep = 1.0 / (1.0 - s.ff_dropout) if (train and s.ff_dropout > 0) else 1.0
        lay.bwd_rec(lw, x, ws['lens_d'], dY, lay.ldy, train, ws['dEpre'].data_ptr(), self.F8,
                    d_in_bf16_mask=(ws['E'].data_ptr(), self.F8), d_in_alpha=keep, **fin)

    def _bwd_aux(self, ws, train):
        s = self.spec
        if ws['use_aux']:
            l = s.aux_layer
            lay, lw = self.enc[l], ws['enc'][l]
            with self.gemm_group():          # (the head's weight gradients leave together, behind its input-gradient chain)
                self.aux.bwd(ws['aux'], lw['Ydrop'].data_ptr(), ws['dP'], ws['dY'][l].data_ptr(), lay.ldy, ws['have_dy'][l], train,
                             d_in_drop=lay.out_drop(train))
            ws['have_dy'][l] = True
        for ax, hx, wx in zip(self.aux_x, s.aux_extra, ws['auxx']):
            if not wx.get('use'):
                continue
            l = hx['layer']
            lay, lw = self.enc[l], ws['enc'][l]
            ax.bwd(wx['ff'], lw['Ydrop'].data_ptr(), wx['dP'], ws['dY'][l].data_ptr(), lay.ldy, ws['have_dy'][l], train,
                   d_in_drop=lay.out_drop(train))
            ws['have_dy'][l] = True

    def _bwd_enc_weights(self, ws, l):
        """Weight gradients of encoder layer l (and, for l == 0, of the subject's conv front-end)."""
        if not self._in_group:
            # every K-major product of the stage (dW_x, dW_h; for the bottom layer also the conv kernels) in one grouped launch
            with self.gemm_group():
                self._bwd_enc_weights(ws, l)
            return
        s, store = self.spec, self.store
        M, Mk = ws['M'], ws['Mk']
        st = self.stream
        x = ws['E'].data_ptr() if l == 0 else ws['enc'][l - 1]['Ydrop'].data_ptr()
        self.enc[l].bwd_weights(ws['enc'][l], x)
        if l > 0:
            return
        # conv front-end weights: dK = A^T . dEpre  (ones row of AT yields the bias gradient)
        sid = ws['sid']
        if len(self.conv[sid]) > 1:
            assert self.tn, 'the conv stack uses the K-major weight-gradient products'
            self._conv_stack_bwd(ws)            # upper conv layers; leaves the bottom layer's pre-activation gradient in ws['dEpre0']
        if ws.get('A_stale'):          # fused forward: the packed bf16 copy is made here, off the forward critical path
            lib.e2t_conv_pack(ws['X'].data_ptr(), ws['lens'].data_ptr(), ws['B'], ws['T'], ws['C'], s.decimation,
                              ws['A'].data_ptr(), ws['Kc8'], st)
            ws['A_stale'] = False
        if self.tn:
            self.gemm(ws['A'].data_ptr(), ws['Kc8'], ws['dEpre0'].data_ptr(), ws['ld0'], store.ptr('conv%s.W' % sid, store.g),
                      ws['out0'], ws['Kc'] + 1, ws['out0'], ws['M0'], splitk=True, tn=True)
            return
        lib.e2t_transpose_bf16(ws['dEpre'].data_ptr(), self.F8, M, s.enc_embed, ws['dEpreT'].data_ptr(), Mk, st)
        lib.e2t_transpose_bf16(ws['A'].data_ptr(), ws['Kc8'], M, ws['Kc'], ws['AT'].data_ptr(), Mk, st)
        self.gemm(ws['AT'].data_ptr(), Mk, ws['dEpreT'].data_ptr(), Mk, store.ptr('conv%s.W' % sid, store.g), s.enc_embed,
                  ws['Kc'] + 1, s.enc_embed, Mk, splitk=True)

    # ------------------------------------------------------------------ saliency (row a12)
    def input_gradient(self, ws):
        """d(loss)/d(encoder_inputs) [B,T,C] fp32 after forward()+backward(): dA = dEpre . W_conv^T, then the
        im2row/time-reversal is undone (restore_and_get_saliencies, reference trainers.py:722-725)."""
        s, sid, dev = self.spec, ws['sid'], self.device
        Kc, Kc8, M, out0, ld0 = ws['Kc'], ws['Kc8'], ws['M0'], ws['out0'], ws['ld0']      # the BOTTOM conv layer's sizes
        if sid not in self.convB:
            self.convB[sid] = _bf(Kc, ld0, device=dev)
        src = getattr(self.store, self._packed or 'p')
        lib.e2t_cast_pack(self.store.ptr('conv%s.W' % sid, src), out0, 1, Kc, out0, self.convB[sid].data_ptr(), ld0, self.stream)
        if 'dA' not in ws:
            ws['dA'] = _f32(M, Kc8, device=dev)
            ws['dX'] = _f32(ws['B'], ws['T'], ws['C'], device=dev)
        self.gemm(ws['dEpre0'].data_ptr(), ld0, self.convB[sid].data_ptr(), ld0, ws['dA'].data_ptr(), Kc8, M, Kc, ld0)
        lib.e2t_conv_unpack_grad_grouped(ws['dA'].data_ptr(), Kc8, ws['lens'].data_ptr(), ws['B'], ws['T'], ws['C'], ws['N0'], ws['G0'],
                                         ws['dX'].data_ptr(), self.stream)
        return ws['dX']

    # ------------------------------------------------------------------ optimiser
    def adam_ranges(self, ranges, step_offset=0):
        """Adam + EMA on [a,b) element ranges of the flat buffers (step_offset=1: before this step's e2t_inc_step)."""
        store = self.store
        st = self.stream
        h = self._adam_hyper(step_offset)
        for a, b in ranges:
            o = 4 * a
            lib.e2t_adam_ema_step(store.p.data_ptr() + o, store.g.data_ptr() + o, store.m.data_ptr() + o,
                                  store.v.data_ptr() + o, store.ema.data_ptr() + o, b - a, self.step_t.data_ptr(),
                                  C.byref(h), st)

    def _adam_hyper(self, step_offset=0):
        h = H.AdamHyper()
        h.lr, h.beta1, h.beta2, h.eps = self.hyper['lr'], self.hyper['beta1'], self.hyper['beta2'], self.hyper['eps']
        h.ema_decay, h.grad_scale, h.step_offset = self.hyper['ema_decay'], self.grad_scale, step_offset
        h.skip_if_nonzero = self.sync_err.data_ptr()      # a step whose in-kernel wait timed out must not reach the weights
        return h

    def adam_step(self, sid=None, repack=True, skip_below=0):
        """Adam + EMA on the shared body and (if given) subject `sid`'s conv; then re-pack operands (repack=False: the
        caller does it at the start of its next forward pass, see forward(pack_first=True)).  skip_below: elements
        [0, skip_below) were already updated by adam_ranges(..., step_offset=1)."""
        store = self.store
        st = self.stream
        self._img_version += 1
        lib.e2t_inc_step(self.step_t.data_ptr(), self.sync_err.data_ptr(), st)
        h = H.AdamHyper()
        h.lr, h.beta1, h.beta2, h.eps = self.hyper['lr'], self.hyper['beta1'], self.hyper['beta2'], self.hyper['eps']
        h.ema_decay, h.grad_scale = self.hyper['ema_decay'], self.grad_scale
        h.skip_if_nonzero = self.sync_err.data_ptr()
        for a, b in self.trainable_ranges(sid):
            a = max(a, skip_below)
            if b <= a:
                continue
            o = 4 * a
            lib.e2t_adam_ema_step(store.p.data_ptr() + o, store.g.data_ptr() + o, store.m.data_ptr() + o,
                                  store.v.data_ptr() + o, store.ema.data_ptr() + o, b - a, self.step_t.data_ptr(),
                                  C.byref(h), st)
        if repack:
            self.pack('p')
        else:
            self._packed = None
            self._img_early = None

    def trainable_ranges(self, sid=None):
        """Contiguous [a,b) element ranges of the flat buffers that receive updates."""
        store = self.store
        names = []
        for nm in store.order:
            if nm.startswith('conv') and (sid is None or not (nm == 'conv%s.W' % sid or re.fullmatch(r'conv%s\.W\d+' % re.escape(str(sid)), nm))):
                continue
            if self.trainable is not None and nm not in self.trainable:
                continue
            names.append(nm)
        ranges = []
        for nm in names:
            a, b = store.seg_range(nm)
            if ranges and ranges[-1][1] == a:
                ranges[-1][1] = b
            else:
                ranges.append([a, b])
        return [tuple(r) for r in ranges]

    # ------------------------------------------------------------------ steps
    def train_step(self, ws, use_graph=True, sync=None):
        """One optimisation step on the batch staged in ws['X'], ws['Y'], ws['auxT'].

        sync: optional parallel.GradSync; each backward stage's gradient ranges are all-reduced
        asynchronously right after the stage is enqueued, and Adam waits for all of them.

        Captured steps are replayed from a stream of the engine's own, ordered behind the caller's stream and joined back
        into it (two event operations per step).  Reason: hipGraphLaunch (ROCm 7.0) picks the streams of a graph's parallel
        branches from a pool with an unchecked scan that skips entries sharing the LAUNCH stream's queue; launched from the
        default stream it can run off the end of that pool once other libraries (RCCL) have created streams in between
        (SIGSEGV inside libamdhip64; scripts/probe_graph_streams.py reproduces it with torch alone: 4 of 96 launches from
        the default stream, 0 of 96 from a side stream)."""
        if use_graph and self.launch_stream_on and torch.cuda.current_stream(self.device) != self.step_stream():
            # (a caller that steps in a loop avoids the two cross-stream hand-offs per step -- ~25 us at cfg2 -- by running the
            #  loop inside `with engine.on_step_stream():`, as SequenceNetwork.fit and bench.py do)
            with self.on_step_stream():
                self._train_step(ws, use_graph, sync)
            return
        self._train_step(ws, use_graph, sync)

    def step_stream(self):
        if self._lstream is None:
            self._lstream = torch.cuda.Stream(device=self.device)
        return self._lstream

    @contextlib.contextmanager
    def on_step_stream(self):
        """Make the engine's own stream current for the duration: ordered behind everything enqueued so far on the caller's
        stream, and joined back into it on exit (also when the body raises)."""
        cur = torch.cuda.current_stream(self.device)
        ls = self.step_stream()
        if cur == ls or not self.launch_stream_on:
            yield
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        ls.wait_event(ev)
        try:
            with torch.cuda.stream(ls):
                yield
        finally:
            ev2 = torch.cuda.Event()
            ev2.record(ls)
            cur.wait_event(ev2)

    def _train_step(self, ws, use_graph, sync):
        self._img_version += 1
        dp = sync is not None and sync.world > 1
        gc = bool(dp and ws.get('global_counts'))          # losses normalised by the global counts: the exchange is a plain sum
        lazy = use_graph and self.overlap      # re-pack inside the (first) graph
        if self._packed != 'p' and not lazy:
            self.pack('p')
        self.grad_scale = (1.0 if gc else sync.grad_scale) if dp else 1.0
        if not use_graph:
            self.forward(ws, train=True, global_counts=gc)
            self.backward(ws, train=True, after_stage=(lambda i, ranges: self._exchange(sync, ranges)) if dp else None)
            if dp:
                sync.allreduce_flag(self.sync_err[0:1])      # a step that one rank must skip is skipped by every rank
                sync.wait()
            self.adam_step(ws['sid'])
            return
        # the captured Adam launches bake in the trainable ranges and the gradient scale
        one = not dp or (getattr(sync, 'capturable', False) and self.overlap and self.options['dp_one_graph'] and not ws['graph'].get('dp_staged'))
        pf = ws.get('prefetch_key') if (ws.get('prefetch') and not ws.get('packed')) else None
        key = ('train_dp' if dp else 'train', gc, tuple(self.trainable_ranges(ws['sid'])), self.grad_scale,
               tuple(sorted(self.hyper.items())), bool(ws.get('packed')), one, id(sync) if (dp and one) else None, pf)   # (captured collectives belong to THAT communicator)
        ws['prefetched'] = False
        g = ws['graph'].get(key)
        if g is None:
            # warm-up launch outside capture (lazy module loading), then capture
            self.forward(ws, train=True, pack_first=lazy, global_counts=gc)
            self.backward(ws, train=True)
            torch.cuda.synchronize(self.device)
            ws['_prefetch_now'] = pf is not None         # (the warm-up passes above left the batch alone: only the CAPTURED forward prefetches)
            try:
                g = self._capture_any(ws, sync, gc, dp, one, lazy, use_graph)
            finally:
                ws['_prefetch_now'] = False
            if g is None:
                return self._train_step(ws, use_graph, sync)
            ws['graph'][key] = g
        # (a replay does not run forward(): an assessment in between may have left the flag off)
        ws['use_aux'] = bool(self.aux and self.spec.aux_scale != 0.0)
        for hx, wx in zip(self.spec.aux_extra, ws['auxx']):
            wx['use'] = hx.get('scale', 1.0) != 0.0
        ws['prefetched'] = pf is not None
        if dp and not one:
            self._replay_staged(ws, g, sync, lazy)
            return
        if g[1] and self._img_early != 'all' and self._img_early != g[1]:
            self.pack('p')           # the graph assumes that the images of ITS early-updated ranges are current
        g[0].replay()
        self._packed = None          # the bottom layer's images are those of the weights BEFORE this step's update
        self._img_early = g[1] if g[1] else None

    def _capture_any(self, ws, sync, gc, dp, one, lazy, use_graph):
        """The captured form of the step for this process layout (None: every rank falls back to the staged schedule -- try again)."""
        if True:
            g = None
            if one and dp:
                # data parallel: the single graph with the collectives as nodes; should the runtime refuse to record a
                # collective on ANY rank, every rank falls back to one graph per stage with the collectives issued between them
                # (the decision is collective: a rank replaying captured collectives and a rank issuing eager ones would not
                # even agree on the order of their all-reduces)
                why = None
                try:
                    if hasattr(sync, 'warm_up'):
                        # every collective shape of the step once OUTSIDE the capture (RCCL's lazy set-up must not run inside one)
                        sync.warm_up([b - a for (_, _, rr) in self.backward_stages(ws) for a, b in rr])
                    g = self._capture_step(ws, sync, gc)
                except RuntimeError as e:
                    why, g = str(e).splitlines()[0][:200], None
                    torch.cuda.synchronize(self.device)
                refused = int(sync.allreduce_numpy(np.array([0 if why is None else 1], np.int32))[0])
                if refused:
                    print('ecog2txt_amd: the data-parallel step could not be captured as one graph on %d rank(s)%s; every rank uses '
                          'one graph per backward stage' % (refused, ' (here: %s)' % why if why else ''))
                    ws['graph']['dp_staged'] = True
                    return None
            elif dp:
                g = self._capture_staged(ws, lazy, gc)
            else:
                g = self._capture_step(ws)
            return g

    @staticmethod
    def _exchange(sync, ranges):
        for a, b in ranges:
            sync.allreduce_range(a, b)

    def _capture_step(self, ws, sync=None, gc=False):
        """The step as ONE graph: forward, backward (weight gradients on the side stream) and the optimiser, with everything
        above the bottom encoder layer updated and re-packed early.  Returns (graph, early-packed ranges).

        sync (a transport whose collectives can be captured: parallel.RcclSync): the data-parallel step is the SAME graph
        plus collective nodes -- the all-reduce of a stage's gradient ranges is recorded on the communicator's stream behind
        the work that completes them (side stream: weight gradients; main stream: the bottom layer's), the maximum of the ranks'
        `sync_err` words behind the last recurrence, and each optimiser launch waits for the collectives issued before it."""
        # parameters whose gradients are final two stages before the end (head, decoder, top encoder layers, an
        # auxiliary head above the bottom layers) are updated on the side stream under the remaining stages
        nl = len(self.enc)
        dp = sync is not None
        early_end, early, packed_early = 0, None, []
        tr = self.trainable_ranges(ws['sid'])
        if nl >= 2 and self.overlap:
            # ONE early update, behind the LAST side stage (the weight gradients of layer 1, queued under the bottom layer's
            # BPTT): everything in front of the bottom layer's segment is final by then.  The optimiser and re-pack kernels
            # are HBM-bound and pair with the MFMA-bound weight gradients of the bottom layer in the step's tail; issued
            # earlier (one update per stage) they sat between the side stream's GEMM launches and pushed the middle layer's
            # weight gradients into that tail, where two GEMM launches then competed (measured: DESIGN.md, appendix)
            hi = self.store.seg_range('enc0.Wx')[0]
            er = [(a, min(b, hi)) for a, b in tr if a < hi]
            if er:
                early_end, packed_early = hi, er

                def early_fn(er=er):
                    if dp:
                        sync.wait_flag()         # the sync_err maximum and, collectives being ordered, every all-reduce issued before it
                                                 # (these ranges'; not the bottom layer's, which follows)
                    if self.options['fused_tail']:
                        self.adam_pack_ranges(er, step_offset=1, slabs=slab_mode[0])     # update + images in one pass over the weight matrices
                    else:
                        self.adam_ranges(er, step_offset=1)
                        self.pack_ranges(er)
                early = (nl, early_fn)
        g1 = torch.cuda.CUDAGraph()
        slab_mode = [False]
        tail = [(max(a, early_end), b) for a, b in tr if b > early_end]
        # Round 6 (option fused_bottom): the bottom layer's ranges go through the fused kernel as well -- at the END of the step, where
        # the separate reduction + update launches used to close it.  Its images are then current when the step ends, and the next
        # step re-packs only the per-subject front-end images (the first pack launch, next to the weight-free start of the
        # front-end).  `packed_all` = every range whose images this graph rebuilds itself: the early ranges + the bottom layer's,
        # WITHOUT the front-end segments (no image of theirs is in the second pack table), so that it is the same set for every
        # participant's graph (BASELINE config 3: one graph per participant, replayed in turn).
        conv_start = min([self.store.seg_range(nm)[0] for nm in self.store.order if nm.startswith('conv')] or [self.store.n])
        fuse_bottom = bool(packed_early) and not dp and self.options['fused_tail'] and self.options['fused_bottom']
        packed_all = [(a, min(b, conv_start)) for a, b in tr if a < conv_start] if fuse_bottom else list(packed_early)
        tail_slabs = [False]
        if packed_early:
            self._pack_subtable(tuple(packed_early))          # descriptor tables are built outside the capture
            if self.options['fused_tail']:
                self._fused_update_plan(tuple(packed_early))
                if fuse_bottom:
                    self._fused_update_plan(tuple(tail))
                if not dp and self.options['fused_reduce']:
                    # single GPU: the weight gradients of the early ranges are never reduced -- the fused kernel sums their split-K
                    # slabs as it reads them.  One eager backward pass in that mode tells where the launcher leaves the slabs
                    # (its decisions depend on shapes only: the capture below repeats them) and allocates the arenas.
                    for keep in ([tuple(packed_early) + tuple(tail), tuple(packed_early)] if fuse_bottom else [tuple(packed_early)]):
                        self._keep_slabs, self._slab_log = keep, []
                        try:
                            self.backward(ws, train=True)
                            torch.cuda.synchronize(self.device)
                            slab_mode[0] = self._fused_update_plan(tuple(packed_early), slab_log=self._slab_log) is not None
                            tail_slabs[0] = slab_mode[0] and len(keep) > len(packed_early) and \
                                self._fused_update_plan(tuple(tail), slab_log=self._slab_log) is not None
                        finally:
                            if not slab_mode[0]:
                                self._keep_slabs = None
                        if tail_slabs[0] or len(keep) == len(packed_early) or not slab_mode[0]:
                            break         # (else: the bottom layer's slabs cannot be named by its descriptors -- probe again, early ranges only)
        probe_sig = self._slab_sig(self._slab_log) if slab_mode[0] else None

        def exchange(ranges):
            for a, b in ranges:
                sync.allreduce_range(a, b)

        def tail_update():
            # the rest (bottom layer, front-end) is updated on the main stream BEFORE it joins the side stream, which is
            # still busy with the early update and its re-pack (cfg2: the step ended 30 us after the side stream did);
            # like the early update it runs on the un-incremented step counter (step_offset = 1)
            if dp:
                sync.join()
            if fuse_bottom:
                self.adam_pack_ranges(tail, step_offset=1, slabs=tail_slabs[0])
            else:
                self.adam_ranges(tail, step_offset=1)
        try:
            if packed_all:
                self._pack_subtable(('skip',) + tuple(packed_all))
            self._slab_log = []
            with capture(g1):
                if dp:
                    sync.attach()
                self.forward(ws, train=True, pack_first=True, pack_skip=packed_all or None, global_counts=gc)
                self.backward(ws, train=True, early=early, before_join=tail_update, exchange=exchange if dp else None,
                              after_last_rec=(lambda: sync.allreduce_flag(self.sync_err[0:1])) if dp else None)
                lib.e2t_inc_step(self.step_t.data_ptr(), self.sync_err.data_ptr(), self.stream)
            diverged = probe_sig is not None and self._slab_sig(self._slab_log) != probe_sig
        finally:
            # (whatever happens between the probe and the end of the capture: no later eager backward() runs without its reductions)
            self._keep_slabs = None
        if diverged:
            # the capture did not leave the slabs where the probe pass did (the descriptors k_adam_pack reads were built from the
            # probe's addresses): that graph must never be replayed -- capture again with the reductions in place
            self._keep_slabs = None
            prev, self.options['fused_reduce'] = self.options['fused_reduce'], False
            try:
                return self._capture_step(ws, sync, gc)
            finally:
                self.options['fused_reduce'] = prev
        if dp:
            sync.pending_ranges = []                  # (tickets recorded during a capture mean nothing outside it)
            sync._flag_pending = False
        self._packed = None
        self._img_early = None
        return (g1, tuple(packed_all))

    def _capture_staged(self, ws, lazy, gc):
        """The data-parallel step as one graph per backward stage for the BPTT chain (main stream) and one per stage for its
        weight-gradient work (side stream): the all-reduce of a stage's ranges is issued behind the graph that completes
        them, the main chain never waits for the side work (same schedule as the single-GPU graph).  Returns
        ([(main graph, side graph or None, gradient ranges)], optimiser graph)."""
        stages = self.backward_stages(ws)
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(device=self.device)
        graphs = []
        for i, (main, side, ranges) in enumerate(stages):
            gm = torch.cuda.CUDAGraph()
            with capture(gm):
                if i == 0:
                    self.forward(ws, train=True, pack_first=lazy, global_counts=gc)
                    ws['have_dy'] = [False] * len(self.enc)
                    self.run_stage(main, side, True)         # the aux head's backward feeds the chain: joined here
                else:
                    main(True)
            gs = None
            if i > 0 and side is not None:
                gs = torch.cuda.CUDAGraph()
                self._on_side = True
                try:
                    with capture(gs):
                        side(True)
                finally:
                    self._on_side = False
            graphs.append((gm, gs, ranges))
        ga = torch.cuda.CUDAGraph()
        with capture(ga):
            self.adam_step(ws['sid'], repack=not lazy)
        return (graphs, ga)

    def _replay_staged(self, ws, g, sync, lazy):
        cur = torch.cuda.current_stream(self.device)
        tr = self.trainable_ranges(ws['sid'])
        follow = bool(lazy and getattr(sync, 'pending_ranges', None) is not None)     # the optimiser follows the exchange range by range

        def update(w, a, b):
            w.wait()                         # the current stream waits for this collective only
            er = [(max(a, x), min(b, y)) for x, y in tr if x < b and y > a]
            if er:
                self.adam_ranges(er, step_offset=1)

        def replay_stages(side_stream, exchange):
            early_done = None
            for i, (gm, gs, ranges) in enumerate(g[0]):
                last_rec = exchange and i == len(g[0]) - 2
                if gs is not None:
                    ev = torch.cuda.Event()
                    ev.record(cur)                   # everything the side work reads was enqueued on the main stream before
                    side_stream.wait_event(ev)
                    with torch.cuda.stream(side_stream):
                        gs.replay()
                        if exchange:
                            self._exchange(sync, ranges)     # the collective orders itself behind the side stream
                gm.replay()
                if last_rec:
                    # the last kernel that can raise sync_err (the bottom layer's BPTT) has been enqueued: the word's maximum over the
                    # ranks makes a step that one rank must skip a step that every rank skips (the replicas cannot drift apart)
                    sync.allreduce_flag(self.sync_err[0:1])
                    if follow and sync.pending_ranges:
                        # Round 6: the updates of every range exchanged so far (all but the bottom layer's) go out HERE, behind the
                        # bottom layer's BPTT and the flag -- in front of the main chain's last graph (the bottom layer's weight
                        # gradients), next to which they run, as the single-GPU graph's early update does.
                        # (They used to be enqueued behind the WHOLE main chain: profiles/r06_dp_timeline_graph_per_stage.txt,
                        # 110 us of optimiser launches exposed behind the last product, 1.757 vs 1.566 ms per cfg2 step on one rank.)
                        # They are HBM-bound and write fp32 masters and optimiser state, which no backward kernel reads.  No update
                        # may read sync_err before its last writer and the maximum over the ranks are done: a step is applied on every
                        # range and every rank, or on none.
                        # Stream: the SIDE stream, behind its last weight-gradient graph.  (A stream of the optimiser's own, so that
                        # each update waits for its own collective only, was measured: HIP dealt it the MAIN stream's hardware queue,
                        # and an update waiting for its all-reduce held up the main chain's last graph behind it -- 2.02 instead of
                        # 1.68 ms per step.  With eager launches the stream -> queue map is not ours to choose.)
                        evm = torch.cuda.Event()
                        evm.record(cur)
                        side_stream.wait_event(evm)
                        with torch.cuda.stream(side_stream):
                            sync.wait_flag()
                            for w, a, b in list(sync.pending_ranges):
                                update(w, a, b)
                        early_done = len(sync.pending_ranges)
                if gs is None and exchange:
                    self._exchange(sync, ranges)
            if not (exchange and early_done is not None):
                join_side_stream(side_stream)
            return early_done

        def join_side_stream(side_stream):
            ev = torch.cuda.Event()
            ev.record(side_stream)
            cur.wait_event(ev)
        if 'side_stream' not in ws['graph']:
            # Which hardware queue the side stream lands on decides how well its GEMMs interleave with the persistent
            # recurrences of the main stream (measured: 2.10 vs 2.31 ms per step between two stream objects of the same
            # process).  HIP deals streams to its hardware queues round-robin, so try a few and keep the fastest; the
            # trial replays run forward + backward only (no exchange, no optimiser: nothing the ranks must agree on).
            cands = [self._wstream] + [torch.cuda.Stream(device=self.device) for _ in range(3)]
            best = None
            for st in cands:
                replay_stages(st, False)
                torch.cuda.synchronize(self.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                for _ in range(3):
                    replay_stages(st, False)
                e1.record(cur)
                torch.cuda.synchronize(self.device)
                t = e0.elapsed_time(e1)
                if best is None or t < best[0]:
                    best = (t, st)
            ws['graph']['side_stream'] = best[1]
        early_done = replay_stages(ws['graph']['side_stream'], True)
        if follow and sync.pending_ranges:
            # what was exchanged behind the early updates (the bottom layer's ranges, completed by the main chain's last graph): only
            # these updates are exposed behind their all-reduce
            if early_done is None:
                sync.wait_flag()
            for w, a, b in list(sync.pending_ranges)[early_done or 0:]:
                update(w, a, b)
            if early_done is not None:
                # (the main stream joins the side stream -- its weight-gradient graphs and the early updates -- only now, behind its own
                #  last update: joined in front of it, that update waited for the early ones, 30 us of exposed tail;  the step counter
                #  moves when every update, which reads it, is done)
                join_side_stream(ws['graph']['side_stream'])
            sync.wait()                          # (all done: clears the lists)
            lib.e2t_inc_step(self.step_t.data_ptr(), self.sync_err.data_ptr(), self.stream)
            self._packed = None
            self._img_early = None
            return
        sync.wait()
        g[1].replay()
        self._img_early = None
        if lazy:
            self._packed = None          # the images are those of the weights BEFORE this step's update

    def check_sync(self, ws=None):
        """Raise if a bounded in-kernel wait of the persistent recurrences gave up since the last check (results since
        then are invalid; the optimiser kernels skipped their updates on the device: e2t_adam_hyper.skip_if_nonzero).
        The exchange buffers / stamp words are reset so that the next launch starts clean.  Costs one device->host read:
        callers use it where they synchronise anyway (losses, assessment, checkpoints, predictions)."""
        if int(self.sync_err[0].item()) == 0:
            return
        info = self.sync_err.cpu().numpy().tolist()
        self.sync_err.zero_()
        for w in ([ws] if ws is not None else list(self._ws.values())):
            for lw in list(w['enc']) + [w['dec']]:
                for k in ('hx', 'dgx', 'counters', 'flagsb', 'flagsbb', 'dgxb'):
                    if k in lw:
                        lw[k].zero_()
        # (data parallel: word 0 is the MAXIMUM of the ranks' words -- every rank raises, none has updated; the other words are this rank's)
        if info[0] == 7:
            raise RuntimeError('persistent BPTT: a recurrent gate gradient was NaN or infinite (results of this step are invalid, '
                               'the weights were not updated: no parameter range, on no rank) %r' % (info,))
        raise RuntimeError('persistent recurrence: an in-kernel wait timed out (data parallel: on this or another rank) '
                           '(results of this step are invalid, the weights were not updated: no parameter range, on no rank) %r' % (info,))

    def saturation_events(self, reset=True):
        """How often the persistent BPTT clipped a recurrent gate gradient at |x| >= 2 in its exchange copy since the last
        call (e2t_lstm_seq_bwd_persistent, err[8]).  0 in healthy training; > 0 means that loss scales / penalties push gate
        gradients three orders of magnitude above their usual size and the recurrent term of BPTT was clipped."""
        n = int(self.sync_err[8].item())
        if reset and n:
            self.sync_err[8] = 0
        return n

    def losses(self, ws):
        v = ws['loss'].cpu().numpy()
        self.check_sync(ws)
        out = dict(decoder=float(v[0]), accuracy=float(v[2]))
        if ws.get('use_aux'):
            out['aux'] = float(v[1])
        out['total'] = self.spec.dec_scale * out['decoder'] + self.spec.aux_scale * out.get('aux', 0.0)
        for j, (hx, wx) in enumerate(zip(self.spec.aux_extra, ws.get('auxx', []))):
            if wx.get('use'):
                out['aux_x%d' % j] = float(wx['loss'].item())
                out['total'] += float(hx.get('scale', 1.0)) * out['aux_x%d' % j]
        return out
