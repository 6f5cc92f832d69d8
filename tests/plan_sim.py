"""CPU interpreter of a launch plan  --  TEST INFRASTRUCTURE (not shipped, not a fallback).

Builds nothing itself: it takes an ``Engine`` whose context was bound to HOST memory (``device='cpu'``: same descriptors,
same packed weight blob the GPU would get, built by the C++ planner in demfi_amd/csrc/ctx.cpp) or a ``Plan`` on CPU
tensors, and *interprets* every ``demfi_conv`` descriptor and pointwise op with torch CPU ops, following the semantics
documented in include/demfi_hip.h.  Purpose: check the host logic (channel maps, chunking, weight repack, output
routing, buffer wiring) against the oracle without a GPU, so that a mismatch on the GPU box isolates the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from demfi_amd import _lib as L
from demfi_amd.engine import SEG_TRUNK, SEG_HEAD, SEG_ITER
from oracle import demfi_oracle as O


class PlanSim:
    def __init__(self, eng):
        self.e = eng
        self.reg = []
        for t in eng.regions():
            self.reg.append((t.data_ptr(), t.numel() * t.element_size(), t))

    # ---- raw memory access through device-pointer arithmetic ----------------------------------------
    def _flat(self, ptr, is_f32):
        for base, nb, t in self.reg:
            if base <= ptr < base + nb:
                dt = torch.float32 if is_f32 else torch.float16
                assert t.dtype == dt or t.dtype == torch.uint8, (t.dtype, dt)
                flat = t.view(-1)
                if t.dtype == torch.uint8:
                    flat = flat.view(dt)
                esz = 4 if is_f32 else 2
                assert (ptr - base) % esz == 0
                return flat, (ptr - base) // esz
        raise KeyError('pointer %x not inside any engine buffer' % ptr)

    def strided(self, v, nch, H, W, b=0):
        flat, off = self._flat(v.ptr, v.is_f32)
        return torch.as_strided(flat, (nch, H, W), (v.sc, v.sy, v.sx), off + b * v.sb)

    # ---- one convolution descriptor ---------------------------------------------------------------------
    def unpack_weights(self, d):
        f32 = d.dtype == L.F32
        cpk, half = (8, 4) if f32 else (16, 8)
        taps = d.kh * d.kw
        nks = [d.chunks[c].nks for c in range(d.n_chunks)]
        n_k = sum(nks) * cpk
        nblk = d.cout_pad // (32 * d.nco)
        flat, off = self._flat(d.wpack, f32)
        per16 = 4 if f32 else 8
        Wp = torch.zeros(d.cout_pad, n_k, taps)
        lanes = torch.arange(64)
        for blk in range(nblk):
            kbase = 0
            for c in range(d.n_chunks):
                vb = blk * d.w_blk_stride + d.chunks[c].w_off
                n = nks[c] * taps * d.nco * 64
                blk_w = flat[off + vb * per16: off + (vb + n) * per16].float().view(taps, nks[c], d.nco, 64, half)
                for s in range(d.nco):
                    co = (blk * d.nco + s) * 32 + (lanes & 31)                      # [64]
                    for ks in range(nks[c]):
                        kp = kbase + ks * cpk + (lanes >> 5)[:, None] * half + torch.arange(half)[None]   # [64,half]
                        Wp[co[:, None].expand(64, half), kp, :] = blk_w[:, ks, s].permute(1, 2, 0)
                kbase += nks[c] * cpk
        return Wp.view(d.cout_pad, n_k, d.kh, d.kw)

    def conv(self, d):
        f32 = d.dtype == L.F32
        esz = 4 if f32 else 2
        cpk = 32 // esz
        outs = []
        for b in range(d.batch):
            xs = []
            for c in range(d.n_chunks):
                ch = d.chunks[c]
                fill = 0
                for pi in range(ch.first_piece, ch.first_piece + ch.n_pieces):
                    p = d.pieces[pi]
                    assert p.lds_ch == fill, 'piece not contiguous'
                    if p.v.ptr is None:
                        xs.append(torch.zeros(p.nch, d.inH, d.inW))
                    else:
                        u = p.up_shift
                        t = self.strided(p.v, p.nch, d.inH >> u, d.inW >> u, b).float()
                        if u:
                            t = t.repeat_interleave(2, 1).repeat_interleave(2, 2)
                        if not f32:
                            t = t.half().float()       # staged in LDS as fp16
                        xs.append(t)
                    fill += p.nch
                assert fill == ch.nks * cpk
            X = torch.cat(xs, 0)[None]
            Wt = self.unpack_weights(d)
            # explicit zero padding: pad_y / pad_x on the top / left, whatever the output size needs on the bottom / right (the
            # kernels zero-fill every out-of-range tap: 2x2 phase filters use pad 1 or 0)
            need_h = (d.H - 1) * d.stride + d.kh - d.inH - d.pad_y
            need_w = (d.W - 1) * d.stride + d.kw - d.inW - d.pad_x
            Xp = F.pad(X, [d.pad_x, max(need_w, 0), d.pad_y, max(need_h, 0)])
            y = F.conv2d(Xp, Wt, None, stride=d.stride)[0][:, :d.H, :d.W]
            assert y.shape[1:] == (d.H, d.W), (y.shape, d.H, d.W)
            bflat, boff = self._flat(d.bias, True)
            y = y + bflat[boff:boff + d.cout_pad].view(-1, 1, 1)
            if d.cout_perm:
                # layers of the persistent kernels: MFMA row r of a 32-cout subtile holds channel
                # (r>>4)*16 + ((r>>2)&1)*8 + ((r>>3)&1)*4 + (r&3) (include/demfi_hip.h); the octet tables describe the channel order
                r = torch.arange(d.cout_pad)
                q = r % 32
                chan = (r // 32) * 32 + (q >> 4) * 16 + ((q >> 2) & 1) * 8 + ((q >> 3) & 1) * 4 + (q & 3)
                yc = torch.empty_like(y)
                yc[chan] = y
                y = yc
            outs.append(y)
        for b, y in enumerate(outs):
            for o in range(d.cout_pad // 8):
                n = d.oct_n[o]
                if n == 0:
                    continue
                sg = d.segs[d.oct_seg[o]]
                c0 = d.oct_ch[o]
                v = y[o * 8:o * 8 + n]

                def sub(view, cc=c0):
                    w = L.View(view.ptr + cc * view.sc * (4 if view.is_f32 else 2), view.sx, view.sy, view.sc, view.sb,
                               view.is_f32, 0)
                    return w
                if sg.res.ptr is not None:
                    r = self.strided(sub(sg.res), n, d.H, d.W, b).float()
                    if sg.mode == L.MODE_STORE:
                        v = _act(v + r, sg.act)
                    elif sg.mode == L.MODE_MUL:
                        v = torch.sigmoid(v) * r
                    else:
                        z = self.strided(sub(sg.aux), n, d.H, d.W, b).float()
                        v = (1 - z) * r + z * torch.tanh(v)
                else:
                    assert sg.mode == L.MODE_STORE
                    v = _act(v, sg.act)
                sc = sg.scale
                dv = sub(sg.dst)
                dst = L.View(dv.ptr + (sg.dy * dv.sy + sg.dx * dv.sx) * (4 if dv.is_f32 else 2), dv.sx * sc, dv.sy * sc,
                             dv.sc, dv.sb, dv.is_f32, 0)
                out = self.strided(dst, n, d.H, d.W, b)
                out.copy_(v.to(out.dtype))
                if d.pack.ptr is not None and o < 4 and d.pack_oct_ch[o] >= 0:
                    # packed copy (demfi_conv.pack): the octet's channels + zeros up to the next multiple of 4, path dtype, NHWC
                    n4 = (n + 3) // 4 * 4
                    pv = L.View(d.pack.ptr + d.pack_oct_ch[o] * 2, d.pack.sx, d.pack.sy, d.pack.sc, d.pack.sb, 0, 0)
                    pk = self.strided(pv, n4, d.H, d.W, b)
                    pk.copy_(torch.cat([v, torch.zeros(n4 - n, d.H, d.W)], 0).to(pk.dtype))

    # ---- whole segments ---------------------------------------------------------------------------------
    def planes(self, ptr, n, H, W):
        return self.strided(L.View(ptr, 1, W, H * W, 0, 1, 0), n, H, W)

    @staticmethod
    def expand(op):
        """A batched point-wise op (demfi_op.bt.nb > 1: one launch for all per-t contexts) as its nb per-context ops."""
        nb = int(op.bt.nb)
        if nb <= 1:
            return [op]
        out = []
        for q in range(nb):
            o = L.Op.from_buffer_copy(bytes(op))
            o.bt.nb = 1
            for name, st in (('a', op.bt.a), ('b', op.bt.b), ('o', op.bt.o)):
                v = getattr(o, name)
                if v.ptr:
                    v.ptr = v.ptr + q * st
            if o.t:
                o.t = o.t + q * op.bt.t
            for i in range(32):
                if o.p[i]:
                    o.p[i] = o.p[i] + q * op.bt.p[i]
            out.append(o)
        return out

    def run(self, ops):
        e = self.e
        H, W = e.H, e.W
        f32 = e.f32
        for op in [x for big in ops for x in self.expand(big)]:
            k = op.kind
            if k == 0:
                self.conv(e.conv_desc(op.conv))
            elif k == 10:
                # fused residual block == its two convolutions in turn, with the intermediate in PRIVATE memory: the fused kernel
                # keeps it in LDS and never touches the plan's scratch buffer, whose memory the workspace arena (round 5) may have
                # handed to a buffer that is alive right now
                d1 = L.Conv.from_buffer_copy(bytes(e.conv_desc(op.conv)))
                d2 = L.Conv.from_buffer_copy(bytes(e.conv_desc(op.nch)))
                sg = d1.segs[d1.sub_seg[0]]
                assert d2.pieces[0].v.ptr == sg.dst.ptr and sg.dst.sx == 64 and sg.dst.sy == d1.W * 64
                tmp = torch.zeros(d1.batch * d1.H * d1.W * 64 + 64, dtype=torch.float32 if d1.dtype == L.F32 else torch.float16)
                self.reg.append((tmp.data_ptr(), tmp.numel() * tmp.element_size(), tmp))
                sg.dst.ptr = tmp.data_ptr()
                sg.dst.sb = d1.H * d1.W * 64
                d2.pieces[0].v.ptr = tmp.data_ptr()
                d2.pieces[0].v.sb = d1.H * d1.W * 64
                self.conv(d1)
                self.conv(d2)
                self.reg.pop()
            elif k == 11:                                                  # round 6: reset gate of a GRU half-step == its convolution
                self.conv(e.conv_desc(op.conv))
            elif k == 12:
                # z + q + blend in one launch == convz (sigmoid) then convq (GRU epilogue) with z in PRIVATE memory: the fused kernel
                # hands z over through LDS and never touches the plan's z buffer, which has no memory of its own under the arena
                dz = L.Conv.from_buffer_copy(bytes(e.conv_desc(op.conv)))
                dq = L.Conv.from_buffer_copy(bytes(e.conv_desc(op.nch)))
                sz, sq = dz.segs[dz.sub_seg[0]], dq.segs[dq.sub_seg[0]]
                assert sq.aux.ptr == sz.dst.ptr and sz.dst.sx == 64 and sz.dst.sy == dz.W * 64
                tmp = torch.zeros(dz.batch * dz.H * dz.W * 64 + 64, dtype=torch.float16)
                self.reg.append((tmp.data_ptr(), tmp.numel() * tmp.element_size(), tmp))
                for v in (sz.dst, sq.aux):
                    v.ptr = tmp.data_ptr()
                    v.sb = dz.H * dz.W * 64
                self.conv(dz)
                self.conv(dq)
                self.reg.pop()
            elif k == 13:                                                  # visualisation extras (DEMFI_HP_EXTRAS)
                if op.conv == 0:
                    a = self.strided(op.a, op.nch, H, W).float()
                    if op.b.ptr:
                        a = a - self.strided(op.b, op.nch, H, W).float()
                    self.planes(op.p[0], 1, H, W).copy_(a.abs().mean(0, keepdim=True))
                elif op.conv == 1:
                    pl = self.planes(op.p[0], 1, H, W)
                    pl -= pl.min()
                    pl /= pl.max()
                else:
                    self.planes(op.p[0], 1, H, W).copy_(1 - self.planes(op.p[1], 1, H, W))
            elif k == 1:                                                   # pack planar fp32 planes -> NHWC slice
                dst = self.strided(op.o, op.nch, H, W)
                for c in range(op.nch):
                    if op.p[c] is None:
                        dst[c] = 0
                    else:
                        dst[c] = self.planes(op.p[c], 1, H, W)[0].to(dst.dtype)
            elif k == 2:                                                   # s2d: x [3,4,H,W] -> [H/2,W/2,48]
                x = self.planes(op.p[0], 12, H, W).view(3, 4, H, W)
                cat = x.permute(1, 0, 2, 3).reshape(1, 12, H, W)
                out = self.strided(L.View(op.p[1], 48, (W // 2) * 48, 1, 0, 1 if f32 else 0, 0), 48, H // 2, W // 2)
                out.copy_(O.space_to_depth(cat, 2)[0].to(out.dtype))
            elif k == 3:                                                   # overlay
                x = self.planes(op.p[0], 12, H, W).view(3, 4, H, W)
                self.planes(op.p[1], 3, H, W).copy_(torch.mean(x[:, 0:2], dim=1))
            elif k == 4:                                                   # fgac gather
                rk = self.strided(op.a, op.nch, H, W).float()[None]
                fl = self.planes(op.p[0], 2, H, W)
                out = self.strided(op.o, op.nch, H, W)
                out.copy_(O.fgac_sample(rk, fl[None])[0].to(out.dtype))
            elif k == 8:                                                   # generalised FGAC window (rr = op.conv, map = op._pad)
                rk = self.strided(op.a, op.nch, H, W).float()[None]
                sk = self.strided(op.b, op.nch, H, W).float()[None]
                fl = self.planes(op.p[0], 2, H, W)
                out = self.strided(op.o, op.nch, H, W)
                out.copy_(O.fgac_window(rk, sk, fl[None], op.conv, 0, op._pad)[0][0].to(out.dtype))
            elif k == 9:                                                   # avg_pool (sr = op.conv)
                a = self.strided(op.a, op.nch, H, W).float()[None]
                out = self.strided(op.o, op.nch, H, W)
                out.copy_(F.avg_pool2d(a, 2 * op.conv + 1, 1, op.conv)[0].to(out.dtype))
            elif k == 5:                                                   # gate blend
                g = self.planes(op.p[0], 1, H, W)
                s_, e_ = self.strided(op.a, op.nch, H, W).float(), self.strided(op.b, op.nch, H, W).float()
                out = self.strided(op.o, op.nch, H, W)
                out.copy_((g * s_ + (1 - g) * e_).to(out.dtype))
            elif k == 6:                                                   # CFR
                t = self.planes(op.t, 1, 1, 1).view(1, 1, 1, 1)
                a, bb = O.cfr_flow_align(self.planes(op.p[0], 2, H, W)[None], self.planes(op.p[1], 2, H, W)[None], t)
                out = self.planes(op.p[3], 4, H, W)
                out[0:2].copy_(a[0])
                out[2:4].copy_(bb[0])
                if op.p[5] is not None:                                    # round 6: the packed record [ft | flow_01, flow_10, logit | 0] for enc1
                    pk = self.strided(L.View(op.p[5], 16, W * 16, 1, 0, 1 if f32 else 0, 0), 16, H, W)
                    pk.copy_(torch.cat([out, self.planes(op.p[0], 2, H, W), self.planes(op.p[1], 2, H, W), self.planes(op.p[4], 1, H, W),
                                        torch.zeros(7, H, W)], 0).to(pk.dtype))
            elif k == 7:                                                   # warp + blend
                t = self.planes(op.t, 1, 1, 1).view(1, 1, 1, 1)
                A = self.strided(op.a, op.nch, H, W).float()[None]
                B = self.strided(op.b, op.nch, H, W).float()[None]
                lg = self.planes(op.p[2], 1, H, W)
                r = O.warp_blend(A, self.planes(op.p[0], 2, H, W)[None], B, self.planes(op.p[1], 2, H, W)[None], lg[None], t)
                out = self.strided(op.o, op.nch, H, W)
                out.copy_(r[0].to(out.dtype))
                if op.p[3] is not None:
                    self.planes(op.p[3], 1, H, W).copy_(torch.sigmoid(lg))
                if op.p[4] is not None:                                    # packed NHWC record [out | fa | fb | occ] for the next conv
                    pk = self.strided(L.View(op.p[4], 8, W * 8, 1, 0, 1 if f32 else 0, 0), 8, H, W)
                    pk.copy_(torch.cat([r[0], self.planes(op.p[0], 2, H, W), self.planes(op.p[1], 2, H, W), torch.sigmoid(lg)], 0).to(pk.dtype))
            else:
                raise AssertionError(k)

    def forward(self, x, t, n):
        e = self.e
        e.x.copy_(x[0])
        e.t_dev.fill_(float(t))
        self.run(e.ops(SEG_TRUNK))
        self.run(e.ops(SEG_HEAD))
        for it in range(n):
            self.run(e.ops(SEG_ITER, it))


    def forward_tb(self, x, ts, n):
        """The batched per-t plan (demfi_forward_tb): context c gets time instant ts[c]; one op list for all contexts."""
        from demfi_amd.engine import SEG_TB_HEAD, SEG_TB_ITER
        e = self.e
        e.x.copy_(x[0])
        for c, t in enumerate(ts):
            e._ctxs[e.trunk][c]['t_dev'].fill_(float(t))
        self.run(e.ops(SEG_TRUNK))
        self.run(e.ops(SEG_TB_HEAD))
        for it in range(n):
            self.run(e.ops(SEG_TB_ITER, it))


def _act(v, act):
    if act == L.ACT_RELU:
        return torch.relu(v)
    if act == L.ACT_TANH:
        return torch.tanh(v)
    if act == L.ACT_SIGMOID:
        return torch.sigmoid(v)
    return v
