"""Layer objects of the device engine: the feed-forward stack (_FFStack: auxiliary heads, vocabulary projection) and the
(bi)directional LSTM layer (_Lstm): operand images (K-contiguous / K-major / MFMA-fragment order), workspaces, and the launch
sequences of their forward, BPTT, input-gradient and weight-gradient stages through the C ABI (include/ecog2txt_hip.h).
Reference stages: _encode_sequences (ecog2txt/trainers.py:821-823), the encoder-target heads (786-799), the decoder RNN and
its projection (513-529).  Host-side plumbing only: every FLOP runs in libecog2txt_hip.so."""
import ctypes as C

import torch

from . import hip_lib as H
from .hip_lib import lib
from .params import ceil_div, r8, rk


def _bf(*shape, device):
    return torch.zeros(*shape, dtype=torch.bfloat16, device=device)


def _f32(*shape, device):
    return torch.zeros(*shape, dtype=torch.float32, device=device)


def _i32(*shape, device):
    return torch.zeros(*shape, dtype=torch.int32, device=device)


class _FFStack:
    """hidden ReLU(+FF dropout) layers then a linear layer stored transposed."""

    def __init__(self, eng, prefix, sizes, in_blocks, in_ld, stream0):
        # in_blocks: [(src_row0, n, dst_k0)] maps dense input features onto the padded K layout
        self.eng, self.prefix, self.sizes, self.in_blocks, self.in_ld, self.stream0 = eng, prefix, sizes, in_blocks, in_ld, stream0
        dev = eng.device
        self.nl = len(sizes) - 1
        self.WT, self.WB = [], []
        for i in range(self.nl):
            kin = in_ld if i == 0 else rk(sizes[i])
            self.WT.append(_bf(sizes[i + 1], kin, device=dev))          # B operand of the forward GEMM
            self.WB.append(_bf(kin, rk(sizes[i + 1]), device=dev))      # B operand of the input-gradient GEMM

    def pack_ops(self, ops, src):
        st = self.eng.store
        for i in range(self.nl):
            last = i == self.nl - 1
            fin, fout = self.sizes[i], self.sizes[i + 1]
            blocks = self.in_blocks if i == 0 else [(0, fin, 0)]
            name = '%s%d.%s' % (self.prefix, i, 'WT' if last else 'W')
            for (r0, n, k0) in blocks:
                if last:      # master [out][in]
                    ops.append(('cast', st.ptr(name, src, r0), fin, 1, fout, n, self.WT[i], k0, 0))
                    ops.append(('cast', st.ptr(name, src, r0), 1, fin, n, fout, self.WB[i], 0, k0))
                else:         # master [in+1][out]
                    ops.append(('cast', st.ptr(name, src, r0 * fout), 1, fout, fout, n, self.WT[i], k0, 0))
                    ops.append(('cast', st.ptr(name, src, r0 * fout), fout, 1, n, fout, self.WB[i], 0, k0))

    def bias_ptr(self, i, src):
        st = self.eng.store
        if i == self.nl - 1:
            return st.ptr('%s%d.b' % (self.prefix, i), src)
        return st.ptr('%s%d.W' % (self.prefix, i), src, self.sizes[i] * self.sizes[i + 1])

    def alloc(self, M):
        dev = self.eng.device
        Mk = rk(M)
        ws = dict(M=M, Mk=Mk, act=[], actT=[], dT=[], dpre=[])
        for i in range(self.nl):
            fin = self.sizes[i]
            if i > 0:
                ws['act'].append(_bf(M, rk(fin), device=dev))            # hidden activation i-1
                if rk(fin) > fin:
                    ws['act'][-1][:, fin] = 1.0                          # ones column for the TN weight gradient (no kernel writes it)
                ws['dpre'].append(_bf(M, rk(fin), device=dev))
            t = _bf(fin + 1, Mk, device=dev)
            t[fin, :M] = 1.0                                             # ones row => bias gradient for free
            ws['actT'].append(t)
            ws['dT'].append(_bf(self.sizes[i + 1], Mk, device=dev))
        ws['out'] = _f32(M, self.sizes[-1], device=dev)
        return ws

    def fwd(self, ws, x_ptr, src, train):
        e = self.eng
        M = ws['M']
        cur, ld = x_ptr, self.in_ld
        for i in range(self.nl):
            last = i == self.nl - 1
            fout = self.sizes[i + 1]
            kin = self.in_ld if i == 0 else rk(self.sizes[i])
            if last:
                e.gemm(cur, ld, self.WT[i].data_ptr(), kin, ws['out'].data_ptr(), fout, M, fout, kin,
                       bias=self.bias_ptr(i, src))
            else:
                o = ws['act'][i]
                e.gemm(cur, ld, self.WT[i].data_ptr(), kin, o.data_ptr(), rk(fout), M, fout, kin,
                       bias=self.bias_ptr(i, src), relu=True, out_bf16=True,
                       drop=(e.spec.ff_dropout if train else 0.0, self.stream0 + i, fout))
                cur, ld = o.data_ptr(), rk(fout)
        return ws['out']

    def bwd_dx(self, ws, d_out, d_in_ptr, d_in_ld, accumulate, train, d_in_drop=None):
        """Input-gradient chain (the critical path): d_out bf16 [M][rk(out)] -> hidden pre-activation gradients
        (ws['dpre'], masked by ReLU/dropout in the GEMM epilogue) -> fp32 gradient of the stack's input."""
        e = self.eng
        M = ws['M']
        d, ldd = d_out.data_ptr(), rk(self.sizes[-1])
        keep = 1.0 / (1.0 - e.spec.ff_dropout) if (train and e.spec.ff_dropout > 0) else 1.0
        for i in range(self.nl - 1, -1, -1):
            fin, fout = self.sizes[i], self.sizes[i + 1]
            kin = self.in_ld if i == 0 else rk(fin)
            if i > 0:
                dp = ws['dpre'][i - 1]
                e.gemm(d, ldd, self.WB[i].data_ptr(), rk(fout), dp.data_ptr(), rk(fin), M, fin, rk(fout),
                       out_bf16=True, alpha=keep, mask_src=(ws['act'][i - 1].data_ptr(), rk(fin)))
                d, ldd = dp.data_ptr(), rk(fin)
            else:
                e.gemm(d, ldd, self.WB[0].data_ptr(), rk(fout), d_in_ptr, d_in_ld, M, kin, rk(fout),
                       accumulate=accumulate, drop=d_in_drop)

    def bwd_dw(self, ws, x_ptr, d_out):
        """Weight (+ bias) gradients from the layer inputs and the gradients bwd_dx left in ws['dpre']: K = M rows of
        K-major operands -> TN GEMM where the input carries its ones column (x[:, fin] == 1), else operand transposes
        + NT GEMM.  Nothing downstream depends on it: the engine queues it on the side stream."""
        e = self.eng
        st = e.store
        M, Mk = ws['M'], ws['Mk']
        for i in range(self.nl - 1, -1, -1):
            last = i == self.nl - 1
            fin, fout = self.sizes[i], self.sizes[i + 1]
            d, ldd = (d_out.data_ptr(), rk(fout)) if last else (ws['dpre'][i].data_ptr(), rk(fout))
            xp, xld = (x_ptr, self.in_ld) if i == 0 else (ws['act'][i - 1].data_ptr(), rk(fin))
            blocks = self.in_blocks if i == 0 else [(0, fin, 0)]
            dense = all(k0 == r0 for (r0, n, k0) in blocks) and xld > fin
            if e.tn and dense and (self.ones_col_set if i == 0 else True):
                if last:      # dW^T = d^T . [x | 1]  [out][in + 1]; the last column is the bias gradient
                    e.gemm(d, ldd, xp, xld, st.ptr('%s%d.WT' % (self.prefix, i), st.g), fin, fout, fin + 1, M, splitk=True,
                           last_col_out=st.ptr('%s%d.b' % (self.prefix, i), st.g), tn=True)
                else:         # [dW; db] = [x | 1]^T . d  [in + 1][out]
                    e.gemm(xp, xld, d, ldd, st.ptr('%s%d.W' % (self.prefix, i), st.g), fout, fin + 1, fout, M, splitk=True, tn=True, ones_last_row=True)
                continue
            # transposes (K-contiguous operands for the NT weight-gradient GEMM)
            lib.e2t_transpose_bf16(d, ldd, M, fout, ws['dT'][i].data_ptr(), Mk, e.stream)
            for (r0, n, k0) in blocks:
                lib.e2t_transpose_bf16(xp + 2 * k0, xld, M, n, ws['actT'][i].data_ptr() + 2 * r0 * Mk, Mk, e.stream)
            if last:
                # dW^T = d^T . x  [out][in]; the ones row of actT makes column `in` the bias gradient
                e.gemm(ws['dT'][i].data_ptr(), Mk, ws['actT'][i].data_ptr(), Mk,
                       st.ptr('%s%d.WT' % (self.prefix, i), st.g), fin, fout, fin + 1, Mk, splitk=True,
                       last_col_out=st.ptr('%s%d.b' % (self.prefix, i), st.g))
            else:
                e.gemm(ws['actT'][i].data_ptr(), Mk, ws['dT'][i].data_ptr(), Mk,
                       st.ptr('%s%d.W' % (self.prefix, i), st.g), fout, fin + 1, fout, Mk, splitk=True)

    def bwd(self, ws, x_ptr, d_out, d_in_ptr, d_in_ld, accumulate, train, d_in_drop=None):
        """d_out: bf16 [M][rk(out)] gradient of the final linear output.  Writes weight grads
        into the store and the input gradient (fp32) into d_in_ptr (d_in_drop: dropout mask of the input, applied in
        the GEMM epilogue)."""
        self.bwd_dx(ws, d_out, d_in_ptr, d_in_ld, accumulate, train, d_in_drop)
        self.bwd_dw(ws, x_ptr, d_out)


class _Lstm:
    """One (bi)directional LSTM layer: operand images + launch helpers."""

    def __init__(self, eng, name, ndir, D, in_blocks, in_ld, Hh, stream):
        self.eng, self.name, self.ndir, self.D, self.in_blocks, self.in_ld, self.H, self.stream = \
            eng, name, ndir, D, in_blocks, in_ld, Hh, stream
        dev = eng.device
        self.H8 = r8(Hh)
        self.ldy = rk(ndir * self.H8 + 1)      # (+1: always room for the ones column of the consumers' TN weight-gradient GEMMs)
        self.N4 = ndir * 4 * Hh
        self.UT, self.KB, self.KB4 = ceil_div(Hh, 16), ceil_div(self.H8, 32), ceil_div(4 * Hh, 32)
        self.WxT = _bf(self.N4, in_ld, device=dev)
        self.WxB = _bf(in_ld, rk(self.N4), device=dev)
        self.WhF = _bf(ndir, 4, self.UT, self.KB, 64, 8, device=dev)
        self.WhB = _bf(ndir, self.UT, self.KB4, 64, 8, device=dev)
        # large hidden sizes (cfg4: H = 1024): the waves of a workgroup hold different weights and share the state through
        # LDS (csrc/lstm_big.hip); operand = fragment image over ALL gate columns of the gate-interleaved master
        self.big = bool(H.load().e2t_lstm_big_ok(Hh)) and self.KB > 26
        self.WhG = _bf(ndir, 4 * Hh // 16, Hh // 32, 64, 8, device=dev) if self.big else None

    def pack_ops(self, ops, src):
        st = self.eng.store
        N4, Hh = self.N4, self.H
        for (r0, n, k0) in self.in_blocks:
            ops.append(('cast', st.ptr(self.name + '.Wx', src, r0 * N4), 1, N4, N4, n, self.WxT, k0, 0))
            ops.append(('cast', st.ptr(self.name + '.Wx', src, r0 * N4), N4, 1, n, N4, self.WxB, 0, k0))
        for d in range(self.ndir):
            base = d * Hh * 4 * Hh
            if (4 * Hh) % 4 == 0 and self.WhF[d].is_contiguous():
                ops.append(('frag4', st.ptr(self.name + '.Wh', src, base), 4, 4 * Hh, Hh, Hh, self.WhF[d, 0]))
            else:
                for g in range(4):
                    ops.append(('frag', st.ptr(self.name + '.Wh', src, base + g), 4, 4 * Hh, Hh, Hh, self.WhF[d, g]))
            ops.append(('frag', st.ptr(self.name + '.Wh', src, base), 4 * Hh, 1, Hh, 4 * Hh, self.WhB[d]))
            if self.big:
                ops.append(('frag', st.ptr(self.name + '.Wh', src, base), 1, 4 * Hh, 4 * Hh, Hh, self.WhG[d]))

    def bias_ptr(self, src):
        return self.eng.store.ptr(self.name + '.Wx', src, self.D * self.N4)

    def persistent_ok(self, B, num_cus):
        """One workgroup per CU for the whole layer and the W_h fragments fit the waves' registers (mirrors the checks
        in e2t_lstm_seq_fwd_persistent): 64-utterance x 16-unit workgroups up to H = 416, 32 x 32 up to H = 832."""
        if self.H % 8 != 0:
            return False
        if self.big:
            return ceil_div(B, 64) * self.ndir * (self.H // 32) <= num_cus
        if self.KB <= 13:        # 64 utterances x 16 units per workgroup
            return ceil_div(B, 64) * self.ndir * self.UT <= num_cus
        return self.KB <= 26 and ceil_div(B, 32) * self.ndir * ceil_div(self.UT, 2) <= num_cus

    def persistent_bwd_ok(self, B, num_cus):
        """Mirrors the checks in e2t_lstm_seq_bwd_persistent: 16-utterance x 64-unit workgroups up to H = 416,
        32 x 32 up to H = 800, one per CU."""
        if self.big:
            return self.H % 128 == 0 and ceil_div(B, 64) * self.ndir * (self.H // 32) <= num_cus
        kq = H.load().e2t_bwd_persist_kq(self.H)
        if kq == 0 or self.H % 8 != 0:
            return False
        RT = ceil_div(B, 16)
        nwg = RT * self.ndir * ceil_div(self.UT, 4) if kq <= 13 else ceil_div(RT, 2) * self.ndir * ceil_div(self.UT, 2)
        return nwg <= num_cus

    def alloc(self, S, B):
        dev = self.eng.device
        M, Mk = S * B, rk(S * B)
        nd, Hh = self.ndir, self.H
        ws = dict(S=S, B=B, M=M, Mk=Mk)
        # input projections of all steps, bf16 (ABI 5; fp32 before: half the bytes for the GEMM to write and the recurrence to read);
        # one slack row: the recurrences fetch a lane's 4 units as 32 B whatever H % 4 is
        ws['Gx'] = _bf(M + 1, self.N4, device=dev)
        ws['Yext'] = _bf((S + 3) * B, self.ldy, device=dev)       # block 0 = initial state, S+1.. = zero slack
        ws['Ydrop'] = _bf(M, self.ldy, device=dev)
        if self.ldy > nd * self.H8:
            ws['Ydrop'][:, nd * self.H8] = 1.0        # ones column for the dW_x of the layer above (no kernel writes it)
        RT, UT = ceil_div(B, 16), ceil_div(Hh, 16)
        ws['Cs'] = _f32(S, nd, RT, UT, 2, 64, 2, device=dev)       # lane-native per-step saves (lstm.hip)
        ws['Gs'] = _bf(S, nd, RT, UT, 2, 64, 8, device=dev)         # gates (i, j, f, o) of two units per 16-B slot, bf16
        ws['dG'] = _bf(M + B, rk(self.N4), device=dev)              # block S = zero slack (rows without successor)
        ws['dGT'] = _bf(self.N4, Mk, device=dev)
        ws['YT'] = _bf(nd, Hh, Mk, device=dev)
        ws['xT'] = _bf(self.D + 1, Mk, device=dev)
        ws['xT'][self.D, :M] = 1.0
        ws['dc_carry'] = _f32(B, nd * Hh, device=dev)
        kq = H.load().e2t_bwd_persist_kq(Hh)
        if kq:      # persistent BPTT: per-cluster stamp state and the in-launch dG exchange (include/ecog2txt_hip.h)
            RT = ceil_div(B, 16)
            nflag = RT * nd * 32 if kq <= 13 else ceil_div(RT, 2) * nd * 128
            ws['counters'] = torch.zeros(nflag + 1, dtype=torch.int32, device=dev)
            ws['dgx'] = _bf(2, nd, RT if kq <= 13 else 2 * ceil_div(RT, 2), 4 * kq, 64, 8, device=dev)
        ws['hx'] = _bf(2 * nd * 4 * ceil_div(B, 64) * self.KB * 64 * 8 + 512, device=dev)     # in-launch h exchange (persistent recurrence)
        if self.big:
            ws['flagsb'] = torch.zeros(ceil_div(B, 64) * nd * 128, dtype=torch.int32, device=dev)
            ws['flagsbb'] = torch.zeros(ceil_div(B, 64) * nd * 128, dtype=torch.int32, device=dev)
            ws['dgxb'] = _bf(2 * nd * 4 * ceil_div(B, 64) * (Hh // 8) * 512, device=dev)       # in-launch dG exchange (big BPTT)
        return ws

    def desc(self, ws, train):
        e = self.eng
        d = H.LstmDesc()
        d.S, d.B, d.H, d.ndir, d.ldy = ws['S'], ws['B'], self.H, self.ndir, self.ldy
        d.forget_bias = e.spec.forget_bias
        d.drop_rate = e.spec.rnn_dropout if train else 0.0
        d.drop_seed, d.drop_step, d.drop_stream = e.seed, e.step_t.data_ptr(), self.stream
        return d

    def out_drop(self, train):
        """(rate, stream, logical ld) of the dropout on this layer's output sequence, for a GEMM epilogue that produces a
        gradient with respect to it: every producer of dY masks its own contribution (the mask is linear), so BPTT does
        not spend a Philox evaluation per step and cell on the critical loop.  None: BPTT masks dY itself (no dropout, or
        the padded column layout differs from the logical one)."""
        rate = self.eng.spec.rnn_dropout if train else 0.0
        # Round 6 experiment (engine option big_bptt_masks, OFF): a LARGE layer (H = 1024, csrc/lstm_big.hip) masks dY in its BPTT, so
        # that the input gradient of the layer above (8704 x 2048 x 8192 at BASELINE config 4) has no dropout in its epilogue and
        # runs on the 256 x 256 instance, whose epilogue is the lean one (csrc/gemm.hip: the full body does not unroll over 32
        # accumulator tiles).  Parity green (tests/test_gpu_fullsize_parity.py cfg4 legs), but the step is no faster: 8.27 / 8.21 ms
        # against 8.17 / 8.19 -- beside the 256 x 256 weight-gradient product of the other branch a second 256 x 256 launch gains
        # nothing (as in round 4: the last round of a 256 x 256 product on 128 x 128 tiles, alone 303 -> 279 us, in the step slower)
        if rate > 0 and self.H8 == self.H and not (self.big and self.eng.options.get('big_bptt_masks', False)):
            return (rate, self.stream, self.ndir * self.H)
        return None

    def fwd_gx(self, ws, x_ptr, src):
        """Input projection of all time steps (no recurrence in it: may run ahead on another stream)."""
        self.eng.gemm(x_ptr, self.in_ld, self.WxT.data_ptr(), self.in_ld, ws['Gx'].data_ptr(), self.N4, ws['M'], self.N4,
                      self.in_ld, bias=self.bias_ptr(src), out_bf16=True, alg=(ws['M'], self.N4, self.D))

    def fwd(self, ws, x_ptr, lens, src, train, c0=None, steps=None, gx_done=False, after_gx=None):
        e = self.eng
        M = ws['M']
        if steps is None:
            if not gx_done:
                self.fwd_gx(ws, x_ptr, src)
            if after_gx is not None:
                after_gx()
            steps = (0, ws['S'])
        if e.persistent_fwd and steps == (0, ws['S']) and self.persistent_ok(ws['B'], e.num_cus):
            # whole sequence in one weight-stationary launch (csrc/lstm.hip: k_lstm_seq_fwd_persist)
            d = self.desc(ws, train)
            if self.big:        # csrc/lstm_big.hip: k_lstm_seq_fwd_big
                lib.e2t_lstm_seq_fwd_big(C.byref(d), ws['Gx'].data_ptr(), self.WhG.data_ptr(), ws['Yext'].data_ptr(),
                                         ws['Ydrop'].data_ptr(), ws['Cs'].data_ptr(), ws['Gs'].data_ptr(), lens.data_ptr(),
                                         c0.data_ptr() if c0 is not None else None, ws['hx'].data_ptr(),
                                         ws['flagsb'].data_ptr(), e.sync_err.data_ptr(), e.num_cus, e.stream)
                return
            lib.e2t_lstm_seq_fwd_persistent(C.byref(d), ws['Gx'].data_ptr(), self.WhF.data_ptr(), ws['Yext'].data_ptr(),
                                            ws['Ydrop'].data_ptr(), ws['Cs'].data_ptr(), ws['Gs'].data_ptr(), lens.data_ptr(),
                                            c0.data_ptr() if c0 is not None else None, ws['hx'].data_ptr(),
                                            e.sync_err.data_ptr(), e.num_cus, e.stream)
            return

        def launch(rb0, nrb, stream):
            d = self.desc(ws, train)
            d.rb_begin, d.rb_count = rb0, nrb
            lib.e2t_lstm_seq_fwd(C.byref(d), ws['Gx'].data_ptr(), self.WhF.data_ptr(), ws['Yext'].data_ptr(),
                                 ws['Ydrop'].data_ptr(), ws['Cs'].data_ptr(), ws['Gs'].data_ptr(), lens.data_ptr(),
                                 c0.data_ptr() if c0 is not None else None, steps[0], steps[1], stream)
        if steps[1] - steps[0] > 1:
            e.run_chains(ws['B'], launch)
        else:
            launch(0, 0, e.stream)

    def bwd_rec(self, ws, x_ptr, lens, dY_ptr, lddy, train, d_in_ptr, d_in_ld, c0=None, dh_final=None, dc_final=None,
                dh0=None, dc0=None, d_in_bf16_mask=None, d_in_alpha=1.0, d_in_accumulate=False, before_d_in=None,
                dy_masked=False, d_in_drop=None):
        """BPTT + input gradient (the critical path of the backward pass).  d_in_bf16_mask=(src_ptr, ld): emit the input
        gradient as bf16 masked by src != 0 (conv ReLU/dropout backward fused into the epilogue).  d_in_ptr=None: BPTT
        only (bwd_d_in() later, e.g. on another stream); before_d_in() runs between the two (a stream join)."""
        e = self.eng
        st = e.store
        M, Mk, S, B = ws['M'], ws['Mk'], ws['S'], ws['B']
        nd, Hh = self.ndir, self.H
        p = lambda t: t.data_ptr() if t is not None else None

        def launch(rb0, nrb, stream):
            d = self.desc(ws, train)
            if dy_masked:
                d.drop_rate = 0.0                 # the producers of dY applied the output dropout mask (out_drop)
            d.rb_begin, d.rb_count = rb0, nrb
            lib.e2t_lstm_seq_bwd(C.byref(d), self.WhB.data_ptr(), ws['dG'].data_ptr(), rk(self.N4), dY_ptr, lddy,
                                 ws['Gs'].data_ptr(), ws['Cs'].data_ptr(), lens.data_ptr(), p(c0), p(dh_final), p(dc_final),
                                 ws['dc_carry'].data_ptr(), p(dh0), p(dc0), stream)
        if e.persistent_bwd and self.big and dh0 is None and self.persistent_bwd_ok(B, e.num_cus):
            d = self.desc(ws, train)            # csrc/lstm_big.hip: k_lstm_seq_bwd_big
            if dy_masked:
                d.drop_rate = 0.0
            lib.e2t_lstm_seq_bwd_big(C.byref(d), self.WhB.data_ptr(), ws['dG'].data_ptr(), rk(self.N4), dY_ptr, lddy,
                                     ws['Gs'].data_ptr(), ws['Cs'].data_ptr(), lens.data_ptr(), p(c0), p(dh_final), p(dc_final),
                                     ws['dgxb'].data_ptr(), ws['flagsbb'].data_ptr(), e.sync_err.data_ptr(), e.num_cus, e.stream)
        elif e.persistent_bwd and not self.big and self.persistent_bwd_ok(B, e.num_cus):
            # whole BPTT sweep in one weight-stationary launch (csrc/lstm.hip: k_lstm_seq_bwd_persist)
            d = self.desc(ws, train)
            if dy_masked:
                d.drop_rate = 0.0
            lib.e2t_lstm_seq_bwd_persistent(C.byref(d), self.WhB.data_ptr(), ws['dG'].data_ptr(), rk(self.N4), dY_ptr, lddy,
                                            ws['Gs'].data_ptr(), ws['Cs'].data_ptr(), lens.data_ptr(), p(c0), p(dh_final),
                                            p(dc_final), p(dh0), p(dc0), ws['dgx'].data_ptr(), ws['counters'].data_ptr(),
                                            e.sync_err.data_ptr(), e.num_cus, e.stream)
        else:
            e.run_chains(B, launch)
        if before_d_in is not None:
            before_d_in()
        if d_in_ptr is not None:
            self.bwd_d_in(ws, d_in_ptr, d_in_ld, d_in_bf16_mask, d_in_alpha, d_in_accumulate, d_in_drop)

    def bwd_d_in(self, ws, d_in_ptr, d_in_ld, d_in_bf16_mask=None, d_in_alpha=1.0, d_in_accumulate=False, d_in_drop=None):
        """Input gradient dG . W_x of the dG that bwd_rec left in ws."""
        e = self.eng
        M = ws['M']
        if d_in_bf16_mask is not None:
            e.gemm(ws['dG'].data_ptr(), rk(self.N4), self.WxB.data_ptr(), rk(self.N4), d_in_ptr, d_in_ld, M, self.D,
                   rk(self.N4), out_bf16=True, alpha=d_in_alpha, mask_src=d_in_bf16_mask, alg=(M, self.D, self.N4))
        else:
            # (a dense input layout needs the D real columns only: in_ld = rk(D + 1) would cost cfg4's 8704 x 2112 x 8192 a ninth
            #  column of 256-wide tiles for 64 padding columns nobody reads)
            dense = all(k0 == r0 for (r0, n, k0) in self.in_blocks)
            e.gemm(ws['dG'].data_ptr(), rk(self.N4), self.WxB.data_ptr(), rk(self.N4), d_in_ptr, d_in_ld, M,
                   self.D if dense else self.in_ld, rk(self.N4), accumulate=d_in_accumulate, drop=d_in_drop, alg=(M, self.D, self.N4))

    def bwd_weights(self, ws, x_ptr):
        """dW_x (+ bias) and dW_h from the dG of bwd_rec.  Nothing downstream of the recurrence depends on it, so the
        engine runs it on a side stream under the next layer's BPTT.  Both products have K = S*B rows of activations
        (x, h_{t-1}) and of their gradients (dG) exactly as the layers wrote them -- K-major -- so they go to the TN
        GEMM directly; the bias gradient comes from a ones column kept at x[:, D] (the forward GEMM's weight image is
        zero there).  Layouts the TN form cannot take (input features not dense, no room for the ones column) fall
        back to operand transposes + NT GEMM."""
        e = self.eng
        st = e.store
        M, Mk, B = ws['M'], ws['Mk'], ws['B']
        nd, Hh = self.ndir, self.H
        dense = all(k0 == r0 for (r0, n, k0) in self.in_blocks) and self.in_ld > self.D
        if e.tn and dense and self.ones_col_set:
            # (inside Seq2SeqEngine.gemm_group() both products -- and the caller's other K-major products of the stage --
            #  leave in one grouped launch)
            e.gemm(x_ptr, self.in_ld, ws['dG'].data_ptr(), rk(self.N4), st.ptr(self.name + '.Wx', st.g), self.N4,
                   self.D + 1, self.N4, M, splitk=True, tn=True, ones_last_row=True)
            # h_{t-1} in processing order: ext block t (forward) / t+2 (backward direction).  Both directions in ONE
            # batched launch: twice the tiles, so half the K splits (slabs, workgroup start-ups) for the same fill
            e.gemm(ws['Yext'].data_ptr(), self.ldy, ws['dG'].data_ptr(), rk(self.N4), st.ptr(self.name + '.Wh', st.g), 4 * Hh,
                   Hh, 4 * Hh, M, splitk=True, tn=True,
                   batch=(nd, 2 * B * self.ldy + self.H8, 4 * Hh, Hh * 4 * Hh) if nd > 1 else None)
            return
        lib.e2t_transpose_bf16(ws['dG'].data_ptr(), rk(self.N4), M, self.N4, ws['dGT'].data_ptr(), Mk, e.stream)
        for (r0, n, k0) in self.in_blocks:
            lib.e2t_transpose_bf16(x_ptr + 2 * k0, self.in_ld, M, n, ws['xT'].data_ptr() + 2 * r0 * Mk, Mk, e.stream)
        e.gemm(ws['xT'].data_ptr(), Mk, ws['dGT'].data_ptr(), Mk, st.ptr(self.name + '.Wx', st.g), self.N4,
               self.D + 1, self.N4, Mk, splitk=True)
        for dd in range(nd):
            # h_{t-1} in processing order: ext block t (forward) / t+2 (backward direction)
            row_off = (2 * B if dd == 1 else 0) * self.ldy
            lib.e2t_transpose_bf16(ws['Yext'].data_ptr() + 2 * (row_off + dd * self.H8), self.ldy, M, Hh,
                                   ws['YT'][dd].data_ptr(), Mk, e.stream)
            e.gemm(ws['YT'][dd].data_ptr(), Mk, ws['dGT'].data_ptr() + 2 * dd * 4 * Hh * Mk, Mk,
                   st.ptr(self.name + '.Wh', st.g, dd * Hh * 4 * Hh), 4 * Hh, Hh, 4 * Hh, Mk, splitk=True)

    def bwd(self, ws, x_ptr, *args, **kw):
        self.bwd_rec(ws, x_ptr, *args, **kw)
        self.bwd_weights(ws, x_ptr)
