"""CPU interpreter of an Engine launch plan  --  TEST INFRASTRUCTURE (not shipped, not a fallback).

Builds nothing itself: it takes an ``Engine`` constructed on CPU tensors (same descriptors, same packed
weight blob the GPU would get) and *interprets* every ``demfi_conv`` descriptor and pointwise op with
torch CPU ops, following the semantics documented in include/demfi_hip.h.  Purpose: check the host logic
(channel maps, chunking, weight repack, output routing, buffer wiring of demfi_amd/engine.py) against the
oracle without a GPU, so that a mismatch on the GPU box isolates the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from demfi_amd import _lib as L
from oracle import demfi_oracle as O


class PlanSim:
    def __init__(self, eng):
        self.e = eng
        self.reg = []
        extra = [getattr(eng, n) for n in ('x', 't_dev', 'weight_blob') if hasattr(eng, n)]
        for t in list(eng._keep) + extra:
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
            y = F.conv2d(X, Wt, None, stride=d.stride, padding=(d.pad_y, d.pad_x))[0]
            assert y.shape[1:] == (d.H, d.W), (y.shape, d.H, d.W)
            bflat, boff = self._flat(d.bias, True)
            y = y + bflat[boff:boff + d.cout_pad].view(-1, 1, 1)
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

    # ---- whole segments ---------------------------------------------------------------------------------
    def run(self, ops):
        e = self.e
        H, W = e.H, e.W
        for op in ops:
            k = op[0]
            if k == 'conv':
                self.conv(e._descs[op[1]])
            elif k == 'pack':
                _, arr, dst, nch = op
                for c in range(nch):
                    if arr[c] is None:
                        dst[0, :, :, c] = 0
                    else:
                        v = L.View(arr[c], 1, W, H * W, 0, 1, 0)
                        dst[0, :, :, c] = self.strided(v, 1, H, W)[0].to(dst.dtype)
            elif k == 's2d':
                x = e.x                                                   # [3,4,H,W] -> frames-major 12 planes
                cat = x.permute(1, 0, 2, 3).reshape(1, 12, H, W)
                e.s2d[0].copy_(O.space_to_depth(cat, 2)[0].permute(1, 2, 0).to(e.dtype))
            elif k == 'overlay':
                e.overlay.copy_(torch.mean(e.x[:, 0:2], dim=1))
            elif k == 'fgac':
                b = op[1]
                rk = e.rk[b].permute(2, 0, 1)[None].float()
                fl = e.ffo[0:2] if b == 0 else e.ffo[2:4]
                e.smp[b].copy_(O.fgac_sample(rk, fl[None])[0].permute(1, 2, 0).to(e.dtype))
            elif k == 'gate':
                b = op[1]
                g = e.gate[b][..., None]
                e.aF[b].copy_((g * e.enc[b].float() + (1 - g) * e.E[b].float()).to(e.dtype))
            elif k == 'cfr':
                t = e.t_dev.view(1, 1, 1, 1)
                a, bb = O.cfr_flow_align(e.ffo[None, 0:2], e.ffo[None, 2:4], t)
                e.ft[0:2].copy_(a[0])
                e.ft[2:4].copy_(bb[0])
            elif k == 'warp_fat':
                _, buf, ba, bb, flows, lbuf, lch, obuf, ob, occ_i = op
                t = e.t_dev.view(1, 1, 1, 1)
                A = buf[ba].permute(2, 0, 1)[None].float()
                B = buf[bb].permute(2, 0, 1)[None].float()
                r = O.warp_blend(A, flows[None, 0:2], B, flows[None, 2:4], lbuf[None, lch:lch + 1], t)
                obuf[ob or 0].copy_(r[0].permute(1, 2, 0).to(e.dtype))
                if occ_i is not None:
                    e.occ[occ_i].copy_(torch.sigmoid(lbuf[lch]))
            elif k == 'warp_thin':
                it = op[1]
                dn = e.delta[it + 1]
                t = e.t_dev.view(1, 1, 1, 1)
                r = O.warp_blend(e.sharp1[None, 0:3], dn[None, 0:2], e.sharp1[None, 3:6], dn[None, 2:4], dn[None, 4:5], t)
                e.stnew.copy_(r[0])
                e.occ[it + 1].copy_(torch.sigmoid(dn[4]))
            else:
                raise AssertionError(k)

    def forward(self, x, t, n):
        e = self.e
        e.x.copy_(x[0])
        e.t_dev.fill_(float(t))
        self.run(e.seg_trunk)
        self.run(e.seg_t_head)
        for it in range(n):
            self.run(e.seg_iter[it])


def _act(v, act):
    if act == L.ACT_RELU:
        return torch.relu(v)
    if act == L.ACT_TANH:
        return torch.tanh(v)
    if act == L.ACT_SIGMOID:
        return torch.sigmoid(v)
    return v
