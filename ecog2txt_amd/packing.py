"""Operand packing of the engine: the bf16 images every kernel reads (K-contiguous and K-major GEMM operands, MFMA fragment
images of the recurrent kernels) are rebuilt from the fp32 masters or the EMA shadows by e2t_pack_batch launches driven by
device-resident descriptor tables.  Mixed into Seq2SeqEngine (engine.py)."""
import ctypes as C

import numpy as np
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
        self._img_version += 1
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

    # ------------------------------------------------------------------ update + images in one pass (e2t_adam_pack_batch)
    def _tile_groups(self, ops):
        """The pack operations `ops`, grouped by the sub-matrix of the masters they read (master orientation: rows of
        contiguous elements): [(key = (offset, R, C, row stride), [(image kind, destination address, ld)])], and the operations
        that cannot go through the tile kernel (alignment)."""
        p0 = self.store.p.data_ptr()
        groups, rest = {}, []
        for op in ops:
            if op[0] == 'cast':
                _, sp, rs, cs, R, Cn, dst, k0, r0 = op
                ld = dst.shape[-1]
                dptr = dst.data_ptr() + 2 * (r0 * ld + k0)
                off = (sp - p0) // 4
                if cs == 1:
                    key, img = (off, R, Cn, rs), (H.TILE_CAST, dptr, ld)
                elif rs == 1:
                    key, img = (off, Cn, R, cs), (H.TILE_CAST_T, dptr, ld)
                else:
                    rest.append(op)
                    continue
                ok = ld % 4 == 0 and dptr % 8 == 0
            else:
                _, sp, ns, ks, Nn, Kk, dst = op
                off = (sp - p0) // 4
                if op[0] == 'frag4':
                    key, img = (off, Kk, 4 * Nn, ks), (H.TILE_FRAG4_KN, dst.data_ptr(), ceil_div(Kk, 32))
                    ok = ns == 4
                elif ks == 1:
                    key, img = (off, Nn, Kk, ns), (H.TILE_FRAG_NK, dst.data_ptr(), ceil_div(Kk, 32))
                    ok = True
                elif ns == 1:
                    key, img = (off, Kk, Nn, ks), (H.TILE_FRAG_KN, dst.data_ptr(), ceil_div(Kk, 32))
                    ok = True
                else:
                    rest.append(op)
                    continue
                ok = ok and dst.data_ptr() % 16 == 0
            off, R, Cc, s0 = key
            if not (ok and off % 4 == 0 and s0 % 4 == 0 and Cc % 4 == 0 and s0 >= Cc):
                rest.append(op)
                continue
            groups.setdefault(key, []).append((img, op))
        out = []
        for key, lst in groups.items():
            for i in range(0, len(lst), H.TILE_IMG_MAX):       # (more images than a descriptor holds: the first chunk updates, the others only pack)
                out.append((key, [im for im, _ in lst[i:i + H.TILE_IMG_MAX]], i == 0))
        return out, rest

    def _tile_table(self, groups, slabs=None):
        """slabs: {descriptor key: (address of the sub-matrix's origin in slab 0, slab stride in elements, split count)}"""
        descs = (H.TileDesc * len(groups))()
        nblk = 0
        for d, (key, imgs, _) in zip(descs, groups):
            off, R, Cc, s0 = key
            d.first_block, d.R, d.C, d.nimg, d.src_off, d.s0 = nblk, R, Cc, len(imgs), off, s0
            if slabs and key in slabs:
                d.gslab, d.gstride, d.gsplits = slabs[key]
            for j, (kind, dptr, ld) in enumerate(imgs):
                d.img[j].dst, d.img[j].kind, d.img[j].ld = dptr, kind, ld
            nblk += ceil_div(R, 64) * ceil_div(Cc, 64)
        raw = bytes(descs)
        return (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device), len(groups), nblk)

    def _fused_update_plan(self, key, slab_log=None):
        """For the element ranges `key` of the flat buffers: (tile table of the sub-matrices that are updated AND re-packed in one
        pass, tile table of further images of those sub-matrices (pack only, behind the update), ranges left to the plain optimiser
        kernel -- biases and whatever no image is made of --, table of the pack operations left to e2t_pack_batch).  Built once,
        outside any stream capture."""
        pkey = (key, 'slabs') if slab_log is not None else key
        if slab_log is None and pkey in self._fused_plans:
            return self._fused_plans[pkey]
        if slab_log is not None:
            # (captured graphs hold the ADDRESS of a plan's descriptor tables: a plan is never replaced by an equal one -- every
            #  capture of the same ranges finds the slabs where the first one did -- and a superseded one is kept alive)
            sig = tuple(sorted((e['off'], e['M'], e['N'], e['batch'], e['slab'], e['splits'], e['stride']) for e in slab_log))
            old = self._fused_plans.get(pkey)
            if old is not None and self._fused_sigs.get(pkey) == sig:
                return old
            if old is not None:
                self._fused_retired.append(old)
            self._fused_sigs[pkey] = sig
        plan = None
        if True:
            if self._pack_table is None:
                self.pack('p')
            ops = [op for op in self._pack_ops[1] if self._op_in(op, key)]
            groups, rest = self._tile_groups(ops)
            # a sub-matrix may be updated by the tile kernel only if it lies inside the ranges and no other sub-matrix touches it
            n = self.store.n
            cover = np.zeros(n, np.uint8)
            inside = np.zeros(n, bool)
            for a, b in key:
                inside[a:b] = True

            def cells(k):
                off, R, Cc, s0 = k
                if s0 == Cc:                       # whole rows: one contiguous range
                    return slice(off, off + R * Cc)
                return (off + np.arange(R, dtype=np.int64)[:, None] * s0 + np.arange(Cc, dtype=np.int64)[None, :]).ravel()
            for k in {g[0] for g in groups}:
                cover[cells(k)] += 1
            upd, pack_only = [], []
            for g in groups:
                c = cells(g[0])
                if g[2] and inside[c].all() and (cover[c] == 1).all():
                    upd.append(g)
                else:
                    pack_only.append(g)
            done = np.zeros(n, bool)
            for g in upd:
                done[cells(g[0])] = True
            left = inside & ~done
            # contiguous runs of what is left (cut where a product that kept its slabs begins or ends: a descriptor reads its
            # gradient from ONE place)
            prods = []
            for e in (slab_log or []):
                for z in range(e['batch']):
                    lo = e['off'] + z * e['M'] * e['N']
                    prods.append((lo, lo + e['M'] * e['N'], e['slab'] + 4 * z * e['splits'] * e['stride'], e['stride'], e['splits'], e['N']))
            # two kept products must never share slab space (each launch has an arena of its own: engine.gemm); should a launcher
            # ever hand out overlapping extents the scheme is off for this capture (reductions as ever) rather than silently wrong
            ext = sorted((pr[2], pr[2] + 4 * pr[4] * pr[3]) for pr in prods)
            if any(ext[i][1] > ext[i + 1][0] for i in range(len(ext) - 1)):
                self._fused_plans.pop(pkey, None)
                return None
            idx = np.flatnonzero(np.diff(np.concatenate([[0], left.view(np.int8), [0]])))
            cuts = sorted({x for lo, hi, *_ in prods for x in (lo, hi)})
            plain = []
            for a, b in zip(idx[0::2].tolist(), idx[1::2].tolist()):
                pts = [a] + [c for c in cuts if a < c < b] + [b]
                plain += list(zip(pts[:-1], pts[1:]))
            # what is left (bias rows and vectors, matrices without a tile image: whole 64-float groups, segments being padded to
            # that) rides along in the SAME launch as image-less descriptors of 64-column rows; only ragged ends stay plain ranges
            if upd:
                keep = []
                for a, b in plain:
                    a4 = -(-a // 4) * 4
                    nrow = (b - a4) // 64
                    if nrow > 0 and a4 == a:
                        upd.append(((a, nrow, 64, 64), [], True))
                        a = a + nrow * 64
                    if b > a:
                        keep.append((a, b))
                plain = keep
            slabs = None
            if slab_log is not None:
                # every descriptor that updates elements of a product whose slabs were kept must read them; one that cannot (it
                # straddles a product's border, its rows are not the product's rows, or elements of such a product are left to the
                # plain kernel) makes the whole scheme unusable for this step: the caller then captures with reductions as ever
                slabs = {}

                def find(off):
                    for pr in prods:
                        if pr[0] <= off < pr[1]:
                            return pr
                    return None
                for k_, _, _ in upd:
                    off, R, Cc, s0 = k_
                    pr = find(off)
                    if pr is None:
                        if any(off < hi and lo < off + (R - 1) * s0 + Cc for lo, hi, *_ in prods):
                            return None
                        continue
                    lo, hi, addr, stride, splits, N = pr
                    if off + (R - 1) * s0 + Cc > hi or not (s0 == N or (s0 == Cc and Cc == 64)):
                        return None
                    slabs[k_] = (addr + 4 * (off - lo), stride, splits)
                for a, b in plain:
                    if any(a < hi and lo < b for lo, hi, *_ in prods):
                        return None
            plan = (self._tile_table(upd, slabs) if upd else None, self._tile_table(pack_only) if pack_only else None, plain,
                    self._pack_descs(rest, self.store.p) if rest else None)
            self._fused_plans[pkey] = plan
        return plan

    def adam_pack_ranges(self, ranges, step_offset=0, slabs=False):
        """Adam + EMA on the element ranges AND the re-pack of every image sourced from them: the tile kernel on the weight
        matrices (one pass: 40 instead of 48 bytes per parameter), the plain optimiser kernel on what is left (biases), the
        plain pack kernel on images the tile kernel does not make.  Same bits as adam_ranges + pack_ranges."""
        upd, pack_only, plain, rest = self._fused_plans[(tuple(ranges), 'slabs')] if slabs else self._fused_update_plan(tuple(ranges))
        st, store = self.stream, self.store
        self._img_version += 1
        if plain:
            self.adam_ranges(plain, step_offset=step_offset)
        if upd:
            h = self._adam_hyper(step_offset)
            lib.e2t_adam_pack_batch(upd[0].data_ptr(), upd[1], upd[2], store.p.data_ptr(), store.g.data_ptr(), store.m.data_ptr(),
                                    store.v.data_ptr(), store.ema.data_ptr(), self.step_t.data_ptr(), C.byref(h), st)
        if pack_only:
            lib.e2t_adam_pack_batch(pack_only[0].data_ptr(), pack_only[1], pack_only[2], store.p.data_ptr(), None, None, None, None, None, None, st)
        if rest:
            lib.e2t_pack_batch(rest[0].data_ptr(), rest[1], rest[2], store.p.data_ptr(), st)

    def _op_in(self, op, ranges):
        off = (op[1] - self.store.p.data_ptr()) // 4
        return any(a <= off < b for a, b in ranges)

    def pack(self, which='p', after_head=None, skip_ranges=None):
        """(Re)build every bf16 operand image from the fp32 masters ('p') or the EMA shadows ('ema'): two launches driven
        by device-resident descriptor tables (built once) -- first the (small) images the front-end needs, the conv
        kernels of all subjects, then everything else; after_head() runs between the two (an event record: the conv GEMM
        of a captured step waits for the first launch only, not for the 80 us of the second)."""
        src = getattr(self.store, which)
        self._img_version += 1
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
