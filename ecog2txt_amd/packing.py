"""Operand packing of the engine: the bf16 images every kernel reads (K-contiguous and K-major GEMM operands, MFMA fragment
images of the recurrent kernels) are rebuilt from the fp32 masters or the EMA shadows by e2t_pack_batch launches driven by
device-resident descriptor tables.  Mixed into Seq2SeqEngine (engine.py)."""
import torch

from . import hip_lib as H
from .hip_lib import lib
from .params import ceil_div, conv_seg


class PackingMixin:
    # ------------------------------------------------------------------ packing
    def pack_ranges(self, ranges):
        """Re-pack (from the masters) exactly the images whose source parameters lie in the element ranges `ranges`: used by
        the captured train step right behind the optimiser update of those ranges, so that the next step starts with
        only the bottom layer's images left to build."""
        tab = self._pack_subtable(tuple(ranges))
        if tab:
            lib.e2t_pack_batch(tab[0].data_ptr(), tab[1], tab[2], self.store.p.data_ptr(), self.stream)

    def _pack_subtable(self, key):
        """Descriptor table of the images sourced from the ranges `key` (built once, OUTSIDE any stream capture: it
        allocates); key ('skip', ranges...) = [head table, table of everything else]."""
        tab = self._pack_sub.get(key)
        if tab is None:
            if self._pack_table is None:
                self.pack('p')
            if key and key[0] == 'skip':
                rest = [op for op in self._pack_ops[1] if not self._op_in(op, key[1:])]
                tab = [self._pack_descs(t, self.store.p) for t in (self._pack_ops[0], rest) if t]
            else:
                ops = [op for op in self._pack_ops[1] if self._op_in(op, key)]
                tab = self._pack_descs(ops, self.store.p) if ops else False
            self._pack_sub[key] = tab
        return tab

    def _op_in(self, op, ranges):
        off = (op[1] - self.store.p.data_ptr()) // 4
        return any(a <= off < b for a, b in ranges)

    def pack(self, which='p', after_head=None, skip_ranges=None):
        """(Re)build every bf16 operand image from the fp32 masters ('p') or the EMA shadows ('ema'): two launches driven
        by device-resident descriptor tables (built once) -- first the (small) images the front-end needs, the conv
        kernels of all subjects, then everything else; after_head() runs between the two (an event record: the conv GEMM
        of a captured step waits for the first launch only, not for the 80 us of the second)."""
        src = getattr(self.store, which)
        if skip_ranges:
            # (a captured train step re-packed these images itself, behind their optimiser update: pack_ranges)
            tab = self._pack_subtable(('skip',) + tuple(skip_ranges))
            for i, (dev, n, nblk) in enumerate(tab):
                lib.e2t_pack_batch(dev.data_ptr(), n, nblk, src.data_ptr(), self.stream)
                if i == 0 and after_head is not None:
                    after_head()
            self._packed = None
            return
        self._img_early = 'all' if which == 'p' else None      # every image is current (masters): any captured step may follow
        if self._pack_table is None:
            st, s = self.store, self.spec
            base = st.p          # offsets are relative, identical for p and ema
            head = []
            for sid, Cc in s.channels.items():
                lays = self.conv[sid]
                ci, co, n = lays[0]
                head.append(('cast', st.ptr(conv_seg(sid, 0), base), 1, co, co, n * ci, self.convT[sid], 0, 0))
                for j in range(1, len(lays)):
                    ci, co, n = lays[j]
                    ldp = self.conv_ld[sid][j - 1]
                    for w in range(n):
                        wsrc = st.ptr(conv_seg(sid, j), base, w * ci * co)
                        head.append(('cast', wsrc, 1, co, co, ci, self.convTj[sid][j], w * ldp, 0))         # [out][w*ld + c]
                        head.append(('cast', wsrc, co, 1, ci, co, self.convBj[sid][j], 0, w * ldp))         # [w*ld + c][out]
            ops = []
            for lay in self.enc:
                lay.pack_ops(ops, base)
            if self.aux:
                self.aux.pack_ops(ops, base)
            for ax in self.aux_x:
                ax.pack_ops(ops, base)
            ops.append(('cast', st.ptr('dec.emb', base), s.dec_embed, 1, s.vocab, s.dec_embed, self.emb, 0, 0))
            self.dec.pack_ops(ops, base)
            self.proj.pack_ops(ops, base)
            self._pack_ops = (head, ops)
            self._pack_table = [self._pack_descs(t, base) for t in (head, ops) if t]
        for i, (dev, n, nblk) in enumerate(self._pack_table):
            lib.e2t_pack_batch(dev.data_ptr(), n, nblk, src.data_ptr(), self.stream)
            if i == 0 and after_head is not None:
                after_head()
        self._packed = which

    def _pack_descs(self, ops, base):
        if True:
            descs = (H.PackDesc * len(ops))()
            nblk = 0
            p0 = base.data_ptr()
            for i, op in enumerate(ops):
                d = descs[i]
                d.first_block = nblk
                if op[0] == 'cast':
                    _, sp, rs, cs, R, Cn, dst, k0, r0 = op
                    ld = dst.shape[-1]
                    tr = rs == 1 and cs != 1                 # source contiguous along the image's rows: tiled transpose
                    off = (sp - p0) // 4
                    dptr = dst.data_ptr() + 2 * (r0 * ld + k0)
                    al = off % 4 == 0 and ld % 4 == 0 and dptr % 8 == 0 and Cn % 4 == 0       # 16-B loads / 8-B stores
                    if tr:
                        kind = 4 if (al and R % 4 == 0 and cs % 4 == 0) else 2
                        units = ceil_div(R, 64) * ceil_div(Cn, 64)
                    else:
                        kind = 3 if (al and cs == 1 and rs % 4 == 0) else 0
                        units = R * ceil_div(Cn, 1024) if kind == 3 else R * ceil_div(Cn, 256)
                    nblk += ceil_div(units, H.PACK_UNITS if kind == 3 else 1)
                    d.kind, d.src_off, d.s0, d.s1, d.d0, d.d1, d.ld = kind, off, rs, cs, R, Cn, ld
                    d.dst = dptr
                else:
                    _, sp, ns, ks, Nn, Kk, dst = op
                    KB = ceil_div(Kk, 32)
                    off = (sp - p0) // 4
                    tiled = ns == 1 and ks % 4 == 0 and off % 4 == 0 and Nn % 4 == 0      # contiguous along n: staged via LDS
                    four = op[0] == 'frag4' and off % 4 == 0                               # gate-interleaved: 4 images, one pass
                    if op[0] == 'frag4' and not four:
                        raise RuntimeError('unaligned gate-interleaved weight segment')
                    d.kind, d.src_off, d.s0, d.s1, d.d0, d.d1, d.ld = (6 if four else 5 if tiled else 1), off, ns, ks, Nn, Kk, KB
                    d.dst = dst.data_ptr()
                    if four:
                        units = ceil_div(Nn, 16) * ceil_div(KB, 2)           # (kinds 5 and 6: two k-blocks per workgroup)
                    else:
                        units = ceil_div(ceil_div(Nn, 16), 4) * ceil_div(KB, 2) if tiled else ceil_div(ceil_div(Nn, 16) * KB, 4)
                    nblk += ceil_div(units, 1 if (tiled or four) else H.PACK_UNITS)
            raw = bytes(descs)
            return (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device), len(ops), nblk)
