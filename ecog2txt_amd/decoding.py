"""Decoding with the trained network (assessment: restore_and_assess, the online predictor): greedy search and beam search,
one decoder step per token through the launch-per-step kernels; the input projection of a token is a row of a table made once per set of weights.  Mixed into Seq2SeqEngine (engine.py)."""
import torch

from .hip_lib import lib
from .params import EOS_ID, PAD_ID, capture, ceil_div, r8, rk
from .layers import _bf, _f32, _i32


class DecodingMixin:
    # ------------------------------------------------------------------ decode
    def greedy_decode(self, ws, which='ema', max_len=None, use_graph=False):
        """beam_width 1 decoding (mocha-1_word_sequence.yaml:31); returns int32 [B, L] token ids.
        use_graph: the whole decode of the batch staged in ws -- encoder + L decoder steps of about six launches each -- is
        captured once per (weights, length, input form) and replayed (the shapes are static: every utterance runs all L steps,
        finished ones emit padding); the eager form launches ~6 L + 20 kernels per batch."""
        s = self.spec
        if self._packed != which:
            self.pack(which)
        if use_graph:
            key = ('greedy', which, max_len, bool(ws.get('packed')))
            g = ws['graph'].get(key)
            if g is None:
                self.greedy_decode(ws, which, max_len)                     # warm-up outside the capture
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with self.on_step_stream():
                    with capture(g):
                        self.greedy_decode(ws, which, max_len)
                g = ws['graph'][key] = (g, ws.get('A_stale'))
            with self.on_step_stream():
                g[0].replay()
            # (a replay runs no host code: the bookkeeping encode() does on the host is restored to what the captured call left --
            #  a replay through the one-pass front-end leaves NO im2row copy of this batch, and a later backward() must know)
            ws['A_stale'] = g[1]
            return ws['hyp']
        src = getattr(self.store, which)
        B, L = ws['B'], ws['L']
        max_len = L if max_len is None else min(max_len, L)
        st = self.stream
        self.encode(ws, src, False)
        lib.e2t_decode_init(ws['done'].data_ptr(), ws['hyp'].data_ptr(), ws['U'].data_ptr(), ws['dlens'].data_ptr(), B, L, EOS_ID, PAD_ID, st)
        dw = ws['dec']
        pw = ws['proj']
        table = self._token_projection_table(src, which)
        if B <= self.small_batch_head_max and self.proj.nl == 1 and self.options['small_batch_head']:
            # ONE or TWO utterances (the online predictor's one, trainers.py:925-949): a step = the recurrence's launch + ONE head
            # launch (e2t_greedy_head_small: projection on the vector units, arg-max, bookkeeping, the next step's input-projection
            # row: 13-14 us) instead of gather + recurrence + a 128-row-tile GEMM for B real rows + arg-max (4.6 + 12.8 + 4.6 us).
            # Measured host to host at cfg2 (profiles/r06_latency_b1.txt): B = 1 0.578 vs 0.620 ms, B = 2 equal, B = 4 0.72 vs 0.66,
            # B = 8 0.90 vs 0.69 -- the kernel takes B <= 8, the engine uses it up to small_batch_head_max = 2
            if self._head_scratch is None:
                self._head_scratch = torch.zeros(2 + 2 * 64 * 8, dtype=torch.int32, device=self.device)
            lib.e2t_gather_rows_u32(table.data_ptr(), ws['U'].data_ptr(), B, B, self.dec.N4 // 2, dw['Gx'].data_ptr(), st)
            pr = self.proj
            for l in range(max_len):
                self.dec.fwd(dw, None, ws['dlens'], src, False, c0=ws['c0'], steps=(l, l + 1))
                more = l + 1 < max_len
                lib.e2t_greedy_head_small(dw['Yext'].data_ptr() + 2 * (l + 1) * B * self.dec.ldy, self.dec.ldy, pr.WT[0].data_ptr(), pr.in_ld,
                                          pr.bias_ptr(0, src), B, s.vocab, s.dec_rnn, l, L, EOS_ID, PAD_ID, ws['done'].data_ptr(),
                                          ws['hyp'].data_ptr(), ws['U'].data_ptr() + 4 * (l + 1) * B if l + 1 < L else None,
                                          table.data_ptr() if more else None, self.dec.N4 // 2,
                                          dw['Gx'].data_ptr() + 2 * (l + 1) * B * self.dec.N4 if more else None,
                                          self._head_scratch.data_ptr(), st)
            return ws['hyp']
        for l in range(max_len):
            # input projection of this step's tokens: rows of the table (embedding and projection of a token are the same for every
            # utterance and step -- decoding applies no dropout: the embedding lookup + a 256-row GEMM per step were two launches)
            lib.e2t_gather_rows_u32(table.data_ptr(), ws['U'].data_ptr() + 4 * l * B, B, B, self.dec.N4 // 2,
                                    dw['Gx'].data_ptr() + 2 * l * B * self.dec.N4, st)
            self.dec.fwd(dw, None, ws['dlens'], src, False, c0=ws['c0'], steps=(l, l + 1))
            # logits for this step: run the projection stack on rows [l*B, (l+1)*B) of the ext array (t+1 block)
            self._proj_rows(ws, src, l)
            nxt = ws['U'].data_ptr() + 4 * (l + 1) * B if l + 1 < L else None
            lib.e2t_greedy_step(pw['out'].data_ptr() + 4 * l * B * s.vocab, s.vocab, B, s.vocab, l, L, EOS_ID, PAD_ID,
                                ws['done'].data_ptr(), ws['hyp'].data_ptr(), nxt, st)
        return ws['hyp']

    def beam_decode(self, ws, beam_width, temperature=1.0, which='ema', max_len=None):
        """Beam search (`beam_width`, mocha-1_word_sequence.yaml:31; `temperature`, :82) over the batch staged in ws: returns
        (int32 [B, L] token ids of the best hypothesis, fp32 [B, W] scores of the final beams).  The encoder runs once on the
        B utterances; the decoder runs on B x W rows (hypothesis w of utterance b = row b*W + w of a second workspace) with
        the launch-per-step kernels, e2t_beam_step picks the survivors and e2t_beam_reorder moves their state.
        beam_width 1 is greedy_decode (identical tokens); oracle: oracle/seq2seq.py beam_decode."""
        s = self.spec
        W = int(beam_width)
        assert 1 <= W <= 16
        if self._packed != which:
            self.pack(which)
        src = getattr(self.store, which)
        B, L = ws['B'], ws['L']
        BW = B * W
        max_len = L if max_len is None else min(max_len, L)
        st = self.stream
        wb = self._beam_workspace(ws['sid'], B, W, L)
        if 'beam' not in wb:
            dev = self.device
            wb['beam'] = dict(rep=torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(W).contiguous(),
                              score=[_f32(BW, device=dev), _f32(BW, device=dev)], done=[_i32(BW, device=dev), _i32(BW, device=dev)],
                              hyp=[_i32(BW, L, device=dev), _i32(BW, L, device=dev)], rowmap=_i32(BW, device=dev),
                              tmp_h=_bf(BW, r8(s.dec_rnn), device=dev), tmp_c=_f32(BW, s.dec_rnn, device=dev),
                              init=torch.tensor([0.0] + [float('-inf')] * (W - 1), dtype=torch.float32, device=dev).repeat(B).contiguous())
        bm = wb['beam']
        self.encode(ws, src, False)
        # every hypothesis starts from the encoder's final state: rows of block 0 of the decoder's ext array and of c0, W times each
        lib.e2t_gather_rows_u32(ws['dec']['Yext'].data_ptr(), bm['rep'].data_ptr(), BW, BW, self.dec.ldy // 2, wb['dec']['Yext'].data_ptr(), st)
        lib.e2t_gather_rows_u32(ws['c0'].data_ptr(), bm['rep'].data_ptr(), BW, BW, s.dec_rnn, wb['c0'].data_ptr(), st)
        bm['score'][0].copy_(bm['init'])                          # beam 0: score 0, the others -inf (W copies of one state)
        lib.e2t_fill_u32(bm['done'][0].data_ptr(), BW, 0, st)
        lib.e2t_fill_u32(bm['hyp'][0].data_ptr(), BW * L, PAD_ID, st)
        lib.e2t_fill_u32(wb['U'].data_ptr(), BW, EOS_ID, st)
        lib.e2t_fill_u32(wb['dlens'].data_ptr(), BW, L, st)
        dw, pw = wb['dec'], wb['proj']
        table = self._token_projection_table(src, which)
        RT, UT = ceil_div(BW, 16), ceil_div(s.dec_rnn, 16)
        cs_step = RT * UT * 2 * 64 * 2                            # floats of one step's lane-native cell save
        cur = 0
        for l in range(max_len):
            lib.e2t_gather_rows_u32(table.data_ptr(), wb['U'].data_ptr() + 4 * l * BW, BW, BW, self.dec.N4 // 2,
                                    dw['Gx'].data_ptr() + 2 * l * BW * self.dec.N4, st)
            self.dec.fwd(dw, None, wb['dlens'], src, False, c0=wb['c0'], steps=(l, l + 1))
            self._proj_rows(wb, src, l)
            nxt = wb['U'].data_ptr() + 4 * (l + 1) * BW if l + 1 < L else None
            lib.e2t_beam_step(pw['out'].data_ptr() + 4 * l * BW * s.vocab, s.vocab, B, W, s.vocab, float(temperature), l, L, EOS_ID, PAD_ID,
                              bm['score'][cur].data_ptr(), bm['done'][cur].data_ptr(), bm['hyp'][cur].data_ptr(),
                              bm['score'][1 - cur].data_ptr(), bm['done'][1 - cur].data_ptr(), bm['hyp'][1 - cur].data_ptr(),
                              bm['rowmap'].data_ptr(), nxt, st)
            cur = 1 - cur
            if l + 1 < max_len:
                lib.e2t_beam_reorder(dw['Yext'].data_ptr() + 2 * (l + 1) * BW * self.dec.ldy, self.dec.ldy,
                                     dw['Cs'].data_ptr() + 4 * l * cs_step, BW, s.dec_rnn, bm['rowmap'].data_ptr(),
                                     bm['tmp_h'].data_ptr(), bm['tmp_c'].data_ptr(), st)
        hyp = bm['hyp'][cur].view(B, W, L)[:, 0, :].contiguous()  # survivors are kept best first
        return hyp, bm['score'][cur].view(B, W)

    def _token_projection_table(self, src, which=None):
        """bf16 [V][4 H_d]: the decoder's input projection of every token, W_x . embedding[v] + b, from the images packed last
        (one 1806-row GEMM per decode call in place of an embedding lookup + a B-row GEMM per step: the same operands and K
        order; bit-identical to the per-step product where the launch plan picks the same tile and K split for M = V as for
        M = B -- it does at the tested shapes, tests/test_gpu_kernels.py -- and within fp32 round-off of one bf16 ulp elsewhere)."""
        s = self.spec
        if self._dec_table is None:
            self._dec_table = _bf(s.vocab, self.dec.N4, device=self.device)
        # (round 6: the table depends on the weights only -- it is made once per set of operand images, not once per call: 12 us of a
        #  0.6-ms single-utterance decode.  _img_version moves whenever an image or a parameter may have changed.)
        key = (which, self._img_version) if which is not None else None
        if torch.cuda.is_current_stream_capturing():
            key = None                 # (a captured decode rebuilds the table at every replay: the graph outlives the weights it was captured with)
        if key is not None and self._dec_table_key == key:
            return self._dec_table
        self._dec_table_key = key
        self.gemm(self.emb.data_ptr(), self.E8, self.dec.WxT.data_ptr(), self.E8, self._dec_table.data_ptr(), self.dec.N4,
                  s.vocab, self.dec.N4, self.E8, bias=self.dec.bias_ptr(src), out_bf16=True)
        return self._dec_table

    def _beam_workspace(self, sid, B, W, L):
        """The decoder-side arrays of B x W hypothesis rows, and nothing else: a full training workspace of B x W rows would
        also hold the inputs, the im2row copy and every encoder layer's activations and gradients W times over (tens of GB at
        beam_width 16), and its cache key could alias a live training workspace of the same row count."""
        key = ('beam', sid, B, W, L)
        wb = self._ws.get(key)
        if wb is None:
            s, dev = self.spec, self.device
            BW = B * W
            Md = L * BW
            wb = dict(sid=sid, B=BW, L=L, Md=Md, enc=[], graph={})
            wb['U'] = _i32(Md, device=dev)
            wb['e'] = _bf(Md, self.E8, device=dev)
            wb['dec'] = self.dec.alloc(L, BW)
            wb['c0'] = _f32(BW, s.dec_rnn, device=dev)
            wb['dlens'] = _i32(BW, device=dev)
            wb['proj'] = self.proj.alloc(Md)
            self._ws[key] = wb
        return wb

    def _proj_rows(self, ws, src, l):
        """Projection stack on decoder step l (un-dropped h_t = ext block l+1)."""
        s = self.spec
        B = ws['B']
        pr, pw = self.proj, ws['proj']
        cur = ws['dec']['Yext'].data_ptr() + 2 * (l + 1) * B * self.dec.ldy
        ld = self.dec.ldy
        for i in range(pr.nl):
            last = i == pr.nl - 1
            fout = pr.sizes[i + 1]
            kin = pr.in_ld if i == 0 else rk(pr.sizes[i])
            if last:
                self.gemm(cur, ld, pr.WT[i].data_ptr(), kin, pw['out'].data_ptr() + 4 * l * B * fout, fout, B, fout, kin,
                          bias=pr.bias_ptr(i, src))
            else:
                o = pw['act'][i].data_ptr() + 2 * l * B * rk(fout)
                self.gemm(cur, ld, pr.WT[i].data_ptr(), kin, o, rk(fout), B, fout, kin, bias=pr.bias_ptr(i, src),
                          relu=True, out_bf16=True)
                cur, ld = o, rk(fout)
