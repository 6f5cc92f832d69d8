"""Launch plan of the DeMFI-Net_rb forward on one MI355X.

Host-side only: this module allocates the HBM-resident activation buffers, repacks the state_dict into
MFMA fragment order (once), builds one ``demfi_conv`` descriptor per convolution call site and records the
launch sequence of the hand-written HIP kernels in ``libdemfi_hip.so``.  PyTorch is used for device memory
and the stream only; every arithmetic op of the path runs in the HIP library (no torch op, no fallback).

The plan follows the data flow of DeMFInet.forward (/root/reference/DeMFInet.py:46-179) but not its
execution shape:
  * every ``torch.cat`` is a multi-piece input of the consuming convolution (no concat buffers);
  * RDB dense blocks grow in place, LFF outputs land directly in the 1152-channel GFF input;
  * PixelShuffle / NN-upsample / tanh / sigmoid / ReLU / residual adds / GRU gate math are epilogues or
    fused loads of the convolution kernel;
  * the t-independent trunk (FF_RDB + FAC-FB, 37 % of the MACs) is a separate segment that a
    caller may run once per input window;
  * Mixer.conv_ref1/2 do not depend on the recursion index and are hoisted out of the boosting loop.
Flows, occlusion logits and 3-channel frames stay fp32 planar ("thin"); features are NHWC in the path dtype.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .spec import HyperParams, layer_table


class _Src:
    __slots__ = ('fat', 'ptr', 'sx', 'sy', 'sc', 'sb', 'is_f32', 'cin', 'up')

    def __init__(self, fat, ptr, sx, sy, sc, sb, is_f32, cin, up=0):
        self.fat, self.ptr, self.sx, self.sy, self.sc, self.sb = fat, ptr, sx, sy, sc, sb
        self.is_f32, self.cin, self.up = is_f32, list(cin), up


class _Dst:
    __slots__ = ('view', 'couts', 'act', 'mode', 'res', 'aux', 'scale', 'dy', 'dx')

    def __init__(self, view, couts, act=L.ACT_NONE, mode=L.MODE_STORE, res=None, aux=None, scale=1, dy=0, dx=0):
        self.view, self.couts, self.act, self.mode = view, list(couts), act, mode
        self.res, self.aux, self.scale, self.dy, self.dx = res, aux, scale, dy, dx


def _view(ptr, sx, sy, sc, sb, is_f32):
    return L.View(ptr, sx, sy, sc, sb, 1 if is_f32 else 0, 0)


_NULL_VIEW = L.View(None, 0, 0, 0, 0, 0, 0)


class Plan:
    """Descriptor builder + launcher shared by the full engine and by kernel-level tests: owns the packed
    weight blob, the descriptor array and a list of launch ops."""

    LDS_BUDGET = 78 * 1024

    def __init__(self, H, W, dtype=torch.float16, device='cuda:0', state_dict=None):
        self.lib = L.load()
        self.H, self.W = H, W
        self.dtype = dtype
        self.f32 = dtype == torch.float32
        self.esz = 4 if self.f32 else 2
        self.dt = L.F32 if self.f32 else L.F16
        self.device = torch.device(device)
        self.sd = {k: v.detach().to('cpu', torch.float32).contiguous() for k, v in (state_dict or {}).items()}
        self._descs = []          # host Conv structs
        self._wblobs = []         # (offset, numpy bytes)
        self._wbytes = 0
        self._keep = []           # tensors referenced only by raw pointers
        self._keep_c = []         # ctypes arrays referenced by launch ops
        self.macs = {}            # name -> MACs per launch (algorithmic, unpadded)

    def _fat(self, h, w, c, batch=1):
        t = torch.zeros((batch, h, w, c), dtype=self.dtype, device=self.device)
        self._keep.append(t)
        return t

    def _thin(self, c, h=None, w=None):
        t = torch.zeros((c, h or self.H, w or self.W), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def fsrc(self, buf, cin0, c0=0, nch=None, b=None, up=0):
        """Input piece from a fat buffer [B,h,w,C]: channels [c0,c0+nch) feed original cin [cin0, cin0+nch).
        b=None keeps the batch stride (batched conv), b=k pins image k."""
        B, h, w, Ct = buf.shape
        nch = Ct - c0 if nch is None else nch
        ptr = buf.data_ptr() + (c0 + (0 if b is None else b * h * w * Ct)) * self.esz
        return _Src(True, ptr, Ct, w * Ct, 1, h * w * Ct if b is None else 0, self.f32, range(cin0, cin0 + nch), up)

    def fsrc_map(self, buf, cin, b=0):
        """Input piece = ALL channels of a fat buffer with an explicit channel -> original-cin list (-1 = unused
        padding channel, gets zero weights)."""
        B, h, w, Ct = buf.shape
        assert len(cin) == Ct
        ptr = buf.data_ptr() + b * h * w * Ct * self.esz
        return _Src(True, ptr, Ct, w * Ct, 1, 0, self.f32, cin, 0)

    def pack_op(self, planes, dst):
        """('pack', ...) launch op: planes = list of [H,W] fp32 plane tensors (None = zero) -> fat buffer dst."""
        Ct = dst.shape[-1]
        planes = list(planes) + [None] * (Ct - len(planes))
        arr = (C.c_void_p * Ct)(*[None if p is None else p.data_ptr() for p in planes])
        self._keep_c.append(arr)
        return ('pack', arr, dst, Ct)

    def tsrc(self, buf, cin, c0=0, nch=None):
        """Input piece from a planar fp32 buffer [C,h,w]; cin = list of original input channels."""
        Ct, h, w = buf.shape
        nch = Ct - c0 if nch is None else nch
        cin = list(cin)
        assert len(cin) == nch
        return _Src(False, buf.data_ptr() + c0 * h * w * 4, 1, w, h * w, 0, True, cin)

    def fview(self, buf, c0=0, b=None):
        B, h, w, Ct = buf.shape
        ptr = buf.data_ptr() + (c0 + (0 if b is None else b * h * w * Ct)) * self.esz
        return _view(ptr, Ct, w * Ct, 1, h * w * Ct if b is None else 0, self.f32)

    def tview(self, buf, c0=0, sb=0):
        Ct, h, w = buf.shape
        return _view(buf.data_ptr() + c0 * h * w * 4, 1, w, h * w, sb, True)

    def activation_bytes(self):
        return sum(t.numel() * t.element_size() for t in self._keep)

    # ------------------------------------------------------------------------------------------------
    # convolution descriptor builder
    # ------------------------------------------------------------------------------------------------
    def conv(self, seg, name, srcs, dsts, H, W, stride=1, batch=1, weight=None, bias=None):
        """Append one convolution launch to segment list ``seg``.  H, W: OUTPUT size."""
        if weight is None:
            weight = self.sd[name + '.weight']
            bias = self.sd[name + '.bias']
        if weight.dim() == 5:
            weight = weight[:, :, 0]
        weight = weight.contiguous()
        cout, cin, kh, kw = weight.shape
        esz = self.esz
        cpk = 32 // esz                                   # channels per k-step
        LH, LW = 7 * stride + kh, 31 * stride + kw
        n_oct = sum(-(-len(ds.couts) // 8) for ds in dsts)
        sub = -(-n_oct // 4)
        nco = sub if sub <= 5 else 4
        # LDS per workgroup = haloed input tile + 2-tap weight ring; keep it under the budget (2 workgroups per CU)
        rec = 128
        # the SepConvGRU layers (1x5 / 5x1 over two 64-channel NHWC pieces) run on their own persistent kernel, which wants
        # the two pieces as two 64-channel chunks whatever the general kernel's LDS budget says
        sep = (esz == 2 and stride == 1 and (kh, kw) in ((1, 5), (5, 1)) and len(srcs) == 2 and cout in (64, 128)
               and all(s.fat and len(s.cin) == 64 and not s.up for s in srcs))
        while not sep and rec > 32 and LH * LW * (rec + 16) + 2 * (rec // 32) * nco * 1024 > self.LDS_BUDGET:
            rec //= 2
        # ---- pack the input pieces into chunks of <= rec bytes (fat pieces first: 16-byte aligned) -------
        order = [s for s in srcs if s.fat] + [s for s in srcs if not s.fat]
        covered = sorted(c for s in srcs for c in s.cin if c >= 0)
        assert covered == list(range(cin)), '%s: inputs cover %d channels, weight has %d' % (name, len(covered), cin)
        d = L.Conv()
        chunks, pieces, cin_map = [], [], []
        cur = dict(first=0, fill=0)

        def close_chunk():
            fill = cur['fill']
            if fill == 0:
                return
            padb = (-fill) % 32
            if padb:
                pieces.append((_NULL_VIEW, padb // esz, fill // esz, 0, 0))
                cin_map.extend([-1] * (padb // esz))
                fill += padb
            chunks.append((cur['first'], len(pieces) - cur['first'], fill // 32))
            cur['first'], cur['fill'] = len(pieces), 0

        for s in order:
            done = 0
            n = len(s.cin)
            while done < n:
                if cur['fill'] >= rec:
                    close_chunk()
                room = (rec - cur['fill']) // esz
                if s.fat:
                    if cur['fill'] % 16:
                        padc = (16 - cur['fill'] % 16) // esz
                        pieces.append((_NULL_VIEW, padc, cur['fill'] // esz, 0, 0))
                        cin_map.extend([-1] * padc)
                        cur['fill'] += padc * esz
                        continue
                    take = min(n - done, room)
                    vec = take * esz // 16
                    if vec == 0:
                        close_chunk()
                        continue
                    vec = 1 << (vec.bit_length() - 1)              # 1, 2, 4, 8 vectors per pixel
                    take = vec * 16 // esz
                else:
                    take = min(n - done, room)
                v = _view(s.ptr + done * s.sc * (4 if s.is_f32 else 2), s.sx, s.sy, s.sc, s.sb, s.is_f32)
                pieces.append((v, take, cur['fill'] // esz, s.up, 1 if s.fat else 0))
                cin_map.extend(s.cin[done:done + take])
                cur['fill'] += take * esz
                done += take
        close_chunk()
        assert len(chunks) <= L.MAX_CHUNKS and len(pieces) <= L.MAX_PIECES, \
            '%s: %d chunks / %d pieces' % (name, len(chunks), len(pieces))
        # ---- output routing -------------------------------------------------------------------------------
        cout_map, octs = [], []
        assert len(dsts) <= L.MAX_SEGS
        for si, ds in enumerate(dsts):
            n = len(ds.couts)
            for o in range(0, n, 8):
                k = min(8, n - o)
                octs.append((si, k, o))
                cout_map.extend(ds.couts[o:o + k] + [-1] * (8 - k))
        assert sorted(c for c in cout_map if c >= 0) == list(range(cout)), '%s: outputs do not cover cout' % name
        assert sub == -(-len(cout_map) // 32)
        cout_pad = -(-sub // nco) * nco * 32
        while len(octs) < cout_pad // 8:
            octs.append((0, 0, 0))
            cout_map.extend([-1] * 8)
        # ---- pack weights / bias ------------------------------------------------------------------------
        cin_arr = np.asarray(cin_map, np.int32)
        nks_arr = np.asarray([c[2] for c in chunks], np.int32)
        cout_arr = np.asarray(cout_map, np.int32)
        wnp = weight.numpy()
        nbytes = C.c_int64(0)
        args = (wnp.ctypes.data, cout, cin, kh, kw, cin_arr.ctypes.data, len(cin_map), nks_arr.ctypes.data, len(chunks),
                cout_arr.ctypes.data, cout_pad, nco, self.dt)
        L.check(self.lib.demfi_pack_conv_weights(*args, None, C.byref(nbytes)), 'pack ' + name)
        packed = np.empty(nbytes.value, np.uint8)
        L.check(self.lib.demfi_pack_conv_weights(*args, packed.ctypes.data, C.byref(nbytes)), 'pack ' + name)
        bnp = np.zeros(cout_pad, np.float32)
        bsrc = bias.numpy()
        for i, c in enumerate(cout_map):
            if c >= 0:
                bnp[i] = bsrc[c]
        w_off = self._add_blob(packed)
        b_off = self._add_blob(bnp.view(np.uint8))
        # ---- fill the descriptor --------------------------------------------------------------------------
        d.dtype, d.H, d.W = self.dt, H, W
        up_any = max([s.up for s in srcs] + [0])
        d.inH = H * stride if stride == 2 else H
        d.inW = W * stride if stride == 2 else W
        d.kh, d.kw, d.stride = kh, kw, stride
        d.pad_y, d.pad_x = (1, 1) if stride == 2 else (kh // 2, kw // 2)
        d.batch, d.cout_pad, d.nco, d.rec_bytes = batch, cout_pad, nco, rec
        d.n_chunks, d.n_pieces, d.n_segs = len(chunks), len(pieces), len(dsts)
        taps = kh * kw
        tot_ks = int(nks_arr.sum())
        d.w_blk_stride = tot_ks * taps * nco * 64
        d.wpack, d.bias = w_off, b_off                   # offsets for now, rebased in _upload()
        woff = 0
        for i, (first, npz, nks) in enumerate(chunks):
            d.chunks[i] = L.Chunk(first, npz, nks, 0, woff)
            woff += nks * taps * nco * 64
        for i, (v, nch, lds_ch, up, fat) in enumerate(pieces):
            d.pieces[i] = L.Piece(v, nch, lds_ch, up, fat)
        for i, ds in enumerate(dsts):
            d.segs[i] = L.Seg(ds.view, ds.res or _NULL_VIEW, ds.aux or _NULL_VIEW, ds.act, ds.mode, ds.scale, ds.dy,
                              ds.dx, 0)
        for i, (si, k, o) in enumerate(octs):
            d.oct_seg[i], d.oct_n[i], d.oct_ch[i] = si, k, o
        for sb in range(L.MAX_OCTS // 4):
            d.sub_seg[sb] = -1
        for sb in range(cout_pad // 32):
            o4 = octs[sb * 4:sb * 4 + 4]
            si = o4[0][0]
            ds = dsts[si]

            def fat_ok(v):
                return v is not None and v.ptr and v.sc == 1 and bool(v.is_f32) == self.f32
            ok = all(o[0] == si and o[1] == 8 and o[2] == o4[0][2] + 8 * j for j, o in enumerate(o4)) and o4[0][2] % 8 == 0
            ok = ok and fat_ok(ds.view) and (ds.res is None or fat_ok(ds.res))
            if ds.mode == L.MODE_GRU:
                ok = ok and fat_ok(ds.aux)
            if ds.mode != L.MODE_STORE:
                ok = ok and ds.res is not None
            if ok:
                d.sub_seg[sb] = si
        d.lw_magic = (0x100000000 + LW - 1) // LW
        self._descs.append(d)
        self.macs[name + '#%d' % len(self._descs)] = cout * cin * taps * H * W * batch
        seg.append(('conv', len(self._descs) - 1, name))
        del up_any

    def _add_blob(self, arr_u8):
        off = self._wbytes
        self._wblobs.append((off, arr_u8))
        self._wbytes = (off + arr_u8.nbytes + 255) & ~255
        return off

    def _upload(self):
        """Weights / biases -> one flat HBM blob (the buffer a multi-GPU launch broadcasts over RCCL), descriptors
        -> one device array."""
        host = np.zeros(self._wbytes, np.uint8)
        for off, a in self._wblobs:
            host[off:off + a.nbytes] = a.reshape(-1)
        self.weight_blob = torch.from_numpy(host).to(self.device)
        self._wblobs = None
        self.rebase_weights()

    def rebase_weights(self):
        base = self.weight_blob.data_ptr()
        if getattr(self, '_rebased', False):
            raise RuntimeError('weights already rebased')
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        for d in self._descs:
            d.wpack = base + (d.wpack or 0)
            d.bias = base + (d.bias or 0)
            d.zero_page = self.zero_page.data_ptr()
        self._rebased = True
        n = len(self._descs)
        sz = C.sizeof(L.Conv)
        raw = bytearray(n * sz)
        for i, d in enumerate(self._descs):
            raw[i * sz:(i + 1) * sz] = bytes(d)
        self.desc_dev = torch.frombuffer(raw, dtype=torch.uint8).clone().to(self.device)
        self._desc_sz = sz


    def launch_conv(self, i, stream, what='conv'):
        L.check(self.lib.demfi_conv2d(C.byref(self._descs[i]), self.desc_dev.data_ptr() + i * self._desc_sz, stream), what)


class Engine(Plan):
    """Buffers + descriptors + launch list for one frame size.  ``dtype`` is torch.float16 or torch.float32."""

    # per-t state: buffers written by the per-t segment + its launch lists.  ``n_ctx`` > 1 builds several independent
    # copies ("contexts") so that different time instants t of one window can run concurrently on different streams
    # (they only share the read-only trunk outputs); use_ctx(c) binds context c to the attributes below.
    _T_ATTRS = ('t_dev', 'cfr_acc', 'ft', 'Ft', 'u1', 'u2', 'u3', 'd0', 'd1', 'd2', 'rF', 'delta', 'occ', 'dec_a', 'dec_t',
                'dec_b', 'sharp1', 'frec', 're1', 'ref_enc', 'de1', 'de2', 'bl1', 'xb', 'zb', 'rh', 'h1', 'fo1', 'stnew',
                'misc16', 'ref32', 'agg3s', 'agg3d', 'delta8', 'g_a', 'g_t', 'g_b', 'finals', 'seg_t_head', 'seg_iter')

    # trunk state: the window input, every buffer the trunk segment writes and its launch list.  ``n_trunk`` > 1 builds
    # several trunk contexts, each with its own set of per-t contexts, so that the trunk of the next window can run while
    # the time instants of the current one are still in flight (WindowRunner.run_windows).
    _TR_ATTRS = ('x', 's2d', 'f1', 'x0', 'grow', 'gffcat', 'g0', 'g1', 'up', 'F01', 'ffo', 'enc_a', 'enc_t', 'enc_b', 'rk',
                 'smp', 'E', 'wg', 'gate', 'aF', 'overlay', 'enc', 'seg_trunk')

    def __init__(self, state_dict, H, W, dtype=torch.float16, device='cuda:0', max_updates=3, hp=None, n_ctx=1, n_trunk=1):
        if H % 8 or W % 8:
            raise ValueError('DeMFI-Net needs H, W multiples of 8 (the harness pads to 32): got %dx%d' % (H, W))
        super().__init__(H, W, dtype, device, state_dict)
        self.hp = hp or HyperParams()
        if self.hp.nf != 64 or self.hp.scale_factor != 2:
            raise NotImplementedError('the HIP path is built for nf=64, scale_factor=2 (the released configuration)')
        self.N = max_updates
        self.table = layer_table(self.hp)
        self._trunks, self._ctxs = [], []              # _ctxs[k][c]: per-t context c reading trunk context k
        for _ in range(max(1, n_trunk)):
            self.seg_trunk = []
            self._alloc_trunk()
            self._build_trunk()
            self._trunks.append({k: getattr(self, k) for k in self._TR_ATTRS})
            ctxs = []
            for _ in range(max(1, n_ctx)):
                self.seg_t_head, self.seg_iter = [], []
                self._alloc_t()
                self._build_t()
                ctxs.append({k: getattr(self, k) for k in self._T_ATTRS})
            self._ctxs.append(ctxs)
        self.use_ctx(0)
        self._upload()

    @property
    def n_ctx(self):
        return len(self._ctxs[0])

    @property
    def n_trunk(self):
        return len(self._trunks)

    @property
    def _ctx(self):
        return self._ctxs[self.trunk]

    def use_ctx(self, c, trunk=None):
        """Bind trunk context ``trunk`` (default: the current one) and its per-t context c to this engine's attributes."""
        if trunk is not None or not hasattr(self, 'trunk'):
            self.trunk = trunk or 0
            self.__dict__.update(self._trunks[self.trunk])
        self.__dict__.update(self._ctxs[self.trunk][c])
        self.ctx = c

    # ------------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------------
    def _alloc_trunk(self):
        H, W, N = self.H, self.W, self.N
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        self.x = torch.zeros((3, 4, H, W), dtype=torch.float32, device=self.device)   # module input, batch 1
        # trunk
        self.s2d = self._fat(H2, W2, 48)
        self.f1 = self._fat(H2, W2, 96)
        self.x0 = self._fat(H2, W2, 96)
        self.grow = self._fat(H2, W2, 128)
        self.gffcat = self._fat(H2, W2, 1152)
        self.g0 = self._fat(H2, W2, 96)
        self.g1 = self._fat(H2, W2, 96)
        self.up = self._fat(H, W, 64)
        self.F01 = self._fat(H, W, 64, 2)
        self.ffo = self._thin(5)                      # flow_01 (2), flow_10 (2), occ_0 logit (1)
        self.enc_a = self._fat(H, W, 64, 2)
        self.enc_t = self._fat(H, W, 64, 2)
        self.enc_b = self._fat(H, W, 64, 2)
        self.rk = self._fat(H, W, 64, 2)
        self.smp = self._fat(H, W, 64, 2)
        self.E = self._fat(H, W, 64, 2)
        self.wg = self._fat(H, W, 64, 2)
        self.gate = self._thin(2)
        self.aF = self._fat(H, W, 64, 2)
        self.overlay = self._thin(3)

    def _alloc_t(self):
        H, W, N = self.H, self.W, self.N
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        self.t_dev = torch.zeros((1,), dtype=torch.float32, device=self.device)
        self._keep.append(self.t_dev)
        self.cfr_acc = torch.zeros((self.lib.demfi_cfr_workspace_bytes(H, W) // 8,), dtype=torch.int64, device=self.device)
        self._keep.append(self.cfr_acc)
        self.ft = self._thin(4)                       # flow_t0, flow_t1
        self.Ft = self._fat(H, W, 64)
        self.u1 = self._fat(H2, W2, 64)
        self.u2 = self._fat(H4, W4, 128)
        self.u3 = self._fat(H8, W8, 256)
        self.d0 = self._fat(H8, W8, 256)
        self.d1 = self._fat(H4, W4, 128)
        self.d2 = self._fat(H2, W2, 64)
        self.rF = self._fat(H, W, 64, 3)              # rF0, rF1, rFt
        self.delta = self._thin(5 * (N + 1)).view(N + 1, 5, H, W)     # (flow_t0, flow_t1, occ logit) per step
        self.occ = self._thin(N + 1)                  # sigmoid(occ logit) per step
        self.dec_a = self._fat(H, W, 64, 3)
        self.dec_t = self._fat(H, W, 64, 3)
        self.dec_b = self._fat(H, W, 64, 3)
        self.sharp1 = self._thin(9)                   # S0p, S1p, Stp
        self.frec = [self._fat(H, W, 64), self._fat(H, W, 64)]
        self.re1 = self._fat(H, W, 32)
        self.ref_enc = self._fat(H, W, 32)
        self.de1 = self._fat(H, W, 32)
        self.de2 = self._fat(H, W, 32)
        self.bl1 = self._fat(H, W, 32)
        self.xb = self._fat(H, W, 64)
        self.zb = self._fat(H, W, 64)
        self.rh = self._fat(H, W, 64)
        self.h1 = self._fat(H, W, 64)
        self.fo1 = self._fat(H, W, 32)
        self.stnew = self._thin(3)
        # planar flows / logits / frames packed to NHWC once, so the consuming convs stage them with vector loads
        self.misc16 = self._fat(H, W, 16)
        self.ref32 = self._fat(H, W, 32)
        self.agg3s = self._fat(H, W, 32)
        self.agg3d = self._fat(H, W, 8)
        self.delta8 = self._fat(H, W, 8)
        self.g_a = self._fat(H, W, 64)
        self.g_t = self._fat(H, W, 64)
        self.g_b = self._fat(H, W, 64)
        self.finals = self._thin(9 * N).view(N, 3, 3, H, W)

    # ------------------------------------------------------------------------------------------------
    # the plan
    # ------------------------------------------------------------------------------------------------
    def _resblocks(self, seg, prefix, n, a, t, b, H, W, batch):
        """x_{k+1} = x_k + conv2(relu(conv1(x_k))) ping-ponging between buffers a and b (t = scratch);
        returns the buffer holding the result."""
        cur, other = a, b
        for i in range(n):
            self.conv(seg, '%s.%d.conv1' % (prefix, i), [self.fsrc(cur, 0)],
                      [_Dst(self.fview(t), range(64), L.ACT_RELU)], H, W, batch=batch)
            self.conv(seg, '%s.%d.conv2' % (prefix, i), [self.fsrc(t, 0)],
                      [_Dst(self.fview(other), range(64), res=self.fview(cur))], H, W, batch=batch)
            cur, other = other, cur
        return cur

    def _x_frames(self, cin0):
        """B0, B1, B-1, B2 as thin pieces of the module input x[3,4,H,W] (frame f, colour c at plane c*4+f)."""
        H, W = self.H, self.W
        out = []
        for f in range(4):
            out.append(_Src(False, self.x.data_ptr() + f * H * W * 4, 1, W, 4 * H * W, 0, True,
                            range(cin0 + 3 * f, cin0 + 3 * f + 3)))
        return out

    def _build_trunk(self):
        H, W, N = self.H, self.W, self.N
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        R, T, S = L.ACT_RELU, L.ACT_TANH, L.ACT_SIGMOID
        D = _Dst
        # ============================ trunk: FF_RDB (DeMFInet.py:233-253) ==================================
        tr = self.seg_trunk
        p = 'FF_RDB_Module.'
        tr.append(('s2d',))
        tr.append(('overlay',))
        self.conv(tr, p + 'SFENet1', [self.fsrc(self.s2d, 0)], [D(self.fview(self.f1), range(96))], H2, W2)
        self.conv(tr, p + 'SFENet2', [self.fsrc(self.f1, 0)], [D(self.fview(self.x0), range(96))], H2, W2)
        for i in range(12):
            xin = (lambda cin0: self.fsrc(self.x0, cin0)) if i == 0 else \
                  (lambda cin0, i=i: self.fsrc(self.gffcat, cin0, 96 * (i - 1), 96))
            xres = self.fview(self.x0) if i == 0 else self.fview(self.gffcat, 96 * (i - 1))
            for c in range(4):
                srcs = [xin(0)] + ([self.fsrc(self.grow, 96, 0, 32 * c)] if c else [])
                self.conv(tr, p + 'RDBs.%d.convs.%d.conv.0' % (i, c), srcs,
                          [D(self.fview(self.grow, 32 * c), range(32), R)], H2, W2)
            self.conv(tr, p + 'RDBs.%d.LFF' % i, [xin(0), self.fsrc(self.grow, 96, 0, 128)],
                      [D(self.fview(self.gffcat, 96 * i), range(96), res=xres)], H2, W2)
        self.conv(tr, p + 'GFF.0', [self.fsrc(self.gffcat, 0)], [D(self.fview(self.g0), range(96))], H2, W2)
        self.conv(tr, p + 'GFF.1', [self.fsrc(self.g0, 0)], [D(self.fview(self.g1), range(96), res=self.fview(self.f1))],
                  H2, W2)
        # UPNet.0 + PixelShuffle(2): out[c, 2h+i, 2w+j] = conv[c*4 + i*2 + j, h, w]
        self.conv(tr, p + 'UPNet.0', [self.fsrc(self.g1, 0)],
                  [D(self.fview(self.up), [c * 4 + i * 2 + j for c in range(64)], scale=2, dy=i, dx=j)
                   for i in range(2) for j in range(2)], H2, W2)
        self.conv(tr, p + 'UPNet.2', [self.fsrc(self.up, 0)],
                  [D(self.fview(self.F01, b=0), range(0, 64), T), D(self.fview(self.F01, b=1), range(64, 128), T),
                   D(self.tview(self.ffo), range(128, 133))], H, W)
        # ============================ trunk: FAC-FB (DeMFInet.py:335-358, 386-452) ==========================
        p = 'FAC_FB_Module.'
        self.conv(tr, p + 'conv_first', [self.fsrc(self.F01, 0)], [D(self.fview(self.enc_a), range(64), R)], H, W, batch=2)
        enc = self._resblocks(tr, p + 'feature_extraction', self.hp.num_ResB_FACFB, self.enc_a, self.enc_t, self.enc_b,
                              H, W, 2)
        self.enc = enc
        names = [p + 'shared_FGAC'] * 2 if self.hp.shared_FGAC_flag else [p + 'FGAC_F1toF0', p + 'FGAC_F0toF1']
        for b in range(2):                  # b = 0: F1 -> F0 with flow_01 ; b = 1: F0 -> F1 with flow_10 (346-349)
            fg = names[b]
            ref, src = 1 - b, b
            self.conv(tr, fg + '.conv_ref_k', [self.fsrc(enc, 0, b=ref)], [D(self.fview(self.rk, b=b), range(64))], H, W)
            tr.append(('fgac', b))
            self.conv(tr, fg + '.fusion', [self.fsrc(self.smp, 0, b=b)], [D(self.fview(self.E, b=b), range(64))], H, W)
            self.conv(tr, fg + '.w_gen', [self.fsrc(enc, 0, b=src), self.fsrc(self.E, 64, b=b)],
                      [D(self.fview(self.wg, b=b), range(64), R)], H, W)
            self.conv(tr, fg + '.w_gen_2', [self.fsrc(self.wg, 0, b=b)], [D(self.tview(self.gate, b), [0], S)], H, W)
            tr.append(('gate', b))

    def _build_t(self):
        H, W, N = self.H, self.W, self.N
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        R, T, S = L.ACT_RELU, L.ACT_TANH, L.ACT_SIGMOID
        D = _Dst
        enc = self.enc
        # ============================ per-t head: CFR, FWB, refinement, D1, Ch_Reducer ======================
        th = self.seg_t_head
        th.append(('cfr',))
        th.append(('warp_fat', self.F01, 0, 1, self.ft, self.ffo, 4, self.Ft, None, None))
        p = 'Refine_Module.'
        # Agg1 = cat[aF0, aF1, Ft, flow_t0, flow_t1, flow_01, flow_10, occ_0_logit] (DeMFInet.py:77)
        th.append(self.pack_op([self.ft[i] for i in range(4)] + [self.ffo[i] for i in range(5)], self.misc16))
        self.conv(th, p + 'enc1', [self.fsrc(self.aF, 0, b=0), self.fsrc(self.aF, 64, b=1), self.fsrc(self.Ft, 128),
                                   self.fsrc_map(self.misc16, list(range(192, 201)) + [-1] * 7)],
                  [D(self.fview(self.u1), range(64), R)], H2, W2, stride=2)
        self.conv(th, p + 'enc2', [self.fsrc(self.u1, 0)], [D(self.fview(self.u2), range(128), R)], H4, W4, stride=2)
        self.conv(th, p + 'enc3', [self.fsrc(self.u2, 0)], [D(self.fview(self.u3), range(256), R)], H8, W8, stride=2)
        self.conv(th, p + 'dec0', [self.fsrc(self.u3, 0)], [D(self.fview(self.d0), range(256), R)], H8, W8)
        self.conv(th, p + 'dec1', [self.fsrc(self.d0, 0, up=1), self.fsrc(self.u2, 256)],
                  [D(self.fview(self.d1), range(128), R)], H4, W4)
        self.conv(th, p + 'dec2', [self.fsrc(self.d1, 0, up=1), self.fsrc(self.u1, 128)],
                  [D(self.fview(self.d2), range(64), R)], H2, W2)
        # + cat[flow_t0, flow_t1, occ_0_logit, aF0, aF1] (78-80), tanh on the feature part (86-87)
        d0 = self.delta[0]
        self.conv(th, p + 'dec3', [self.fsrc(self.d2, 0, up=1)],
                  [D(self.fview(self.rF, b=0), range(5, 69), T, res=self.fview(self.aF, b=0)),
                   D(self.fview(self.rF, b=1), range(69, 133), T, res=self.fview(self.aF, b=1)),
                   D(self.tview(d0), range(0, 4), res=self.tview(self.ft)),
                   D(self.tview(d0, 4), [4], res=self.tview(self.ffo, 4))], H, W)
        th.append(('warp_fat', self.rF, 0, 1, d0, d0, 4, self.rF, 2, 0))     # rFt -> rF[2], occ[0]
        # D1 on the three frames (Conv3d depth = batch), DeMFInet.py:95-101
        self.conv(th, 'Dec_first', [self.fsrc(self.rF, 0)], [D(self.fview(self.dec_a), range(64), R)], H, W, batch=3)
        cur = self._resblocks(th, 'Decoder_res', self.hp.num_ResB_Dec, self.dec_a, self.dec_t, self.dec_b, H, W, 3)
        self.conv(th, 'Dec_last1', [self.fsrc(cur, 0)], [D(self.fview(self.dec_t), range(64), R)], H, W, batch=3)
        self.conv(th, 'Dec_last2', [self.fsrc(self.dec_t, 0)], [D(self.tview(self.sharp1, 0, sb=3 * H * W), range(3))],
                  H, W, batch=3)
        self.conv(th, 'Ch_Reducer', [self.fsrc(self.rF, 0, b=0), self.fsrc(self.rF, 64, b=1), self.fsrc(self.rF, 128, b=2)],
                  [D(self.fview(self.frec[0]), range(64), T)], H, W)
        # Mixer reference branch (iteration-invariant, hoisted): cat[S0p,S1p,Stp,B0,B1,B-1,B2 | flow_10,flow_01 | t_ref]
        p = 'Booster_Module.'
        xpl = [self.x[c, f] for f in range(4) for c in range(3)]          # B0, B1, B-1, B2 colour planes (cat order)
        th.append(self.pack_op([self.sharp1[i] for i in range(9)] + xpl +
                               [self.ffo[2], self.ffo[3], self.ffo[0], self.ffo[1]] + [d0[i] for i in range(5)], self.ref32))
        self.conv(th, p + 'Mixer.conv_ref1', [self.fsrc_map(self.ref32, list(range(30)) + [-1, -1])],
                  [D(self.fview(self.re1), range(32), R)], H, W)
        # iteration-invariant part of Agg3 (DeMFInet.py:151-155): S0p,S1p | occ_0 | rflow_t0,t1 | flow_10,flow_01 | frames
        th.append(self.pack_op([self.sharp1[i] for i in range(6)] + [self.occ[0]] + [d0[i] for i in range(4)] +
                               [self.ffo[2], self.ffo[3], self.ffo[0], self.ffo[1]] + xpl, self.agg3s))
        agg3s_cin = list(range(0, 6)) + [73] + list(range(74, 78)) + [78, 79, 80, 81] + list(range(87, 99)) + [-1] * 5
        self.conv(th, p + 'Mixer.conv_ref2', [self.fsrc(self.re1, 0)], [D(self.fview(self.ref_enc), range(32), R)], H, W)
        # ============================ recursive boosting, one list per iteration ============================
        zr = {}
        for s in ('1', '2'):
            zr[s] = (torch.cat([self.sd[p + 'GB.convz' + s + '.weight'], self.sd[p + 'GB.convr' + s + '.weight']], 0),
                     torch.cat([self.sd[p + 'GB.convz' + s + '.bias'], self.sd[p + 'GB.convr' + s + '.bias']], 0))
        for it in range(N):
            sg = []
            self.seg_iter.append(sg)
            dc, dn = self.delta[it], self.delta[it + 1]
            hin, hout = self.frec[it % 2], self.frec[(it + 1) % 2]
            sg.append(self.pack_op([dc[i] for i in range(5)], self.delta8))
            self.conv(sg, p + 'Mixer.conv_delta1', [self.fsrc_map(self.delta8, list(range(5)) + [-1] * 3)],
                      [D(self.fview(self.de1), range(32), R)], H, W)
            self.conv(sg, p + 'Mixer.conv_delta2', [self.fsrc(self.de1, 0)], [D(self.fview(self.de2), range(32), R)], H, W)
            self.conv(sg, p + 'Mixer.conv_blend1', [self.fsrc(self.ref_enc, 0), self.fsrc(self.de2, 32)],
                      [D(self.fview(self.bl1), range(32), R)], H, W)
            self.conv(sg, p + 'Mixer.conv_blend2', [self.fsrc(self.bl1, 0)], [D(self.fview(self.xb), range(64), R)], H, W)
            # SepConvGRU (838-857): z | r share their input -> one 128-cout conv; r*h and the state update are epilogues
            h = hin
            for s, hnext in (('1', self.h1), ('2', hout)):
                self.conv(sg, p + 'GB.convzr' + s, [self.fsrc(h, 0), self.fsrc(self.xb, 64)],
                          [D(self.fview(self.zb), range(0, 64), S),
                           D(self.fview(self.rh), range(64, 128), mode=L.MODE_MUL, res=self.fview(h))],
                          H, W, weight=zr[s][0], bias=zr[s][1])
                self.conv(sg, p + 'GB.convq' + s, [self.fsrc(self.rh, 0), self.fsrc(self.xb, 64)],
                          [D(self.fview(hnext), range(64), mode=L.MODE_GRU, res=self.fview(h), aux=self.fview(self.zb))],
                          H, W)
                h = hnext
            self.conv(sg, p + 'flow_occ.conv1', [self.fsrc(hout, 0)], [D(self.fview(self.fo1), range(32), R)], H, W)
            self.conv(sg, p + 'flow_occ.conv2', [self.fsrc(self.fo1, 0)], [D(self.tview(dn), range(5), res=self.tview(dc))],
                      H, W)
            sg.append(('warp_thin', it))
            # Agg3 (DeMFInet.py:151-155)
            sg.append(self.pack_op([self.stnew[i] for i in range(3)] + [dn[i] for i in range(4)] + [self.occ[it + 1]],
                                   self.agg3d))
            self.conv(sg, 'Dec_first_2',
                      [self.fsrc(hout, 9), self.fsrc_map(self.agg3s, agg3s_cin),
                       self.fsrc_map(self.agg3d, [6, 7, 8, 82, 83, 84, 85, 86])],
                      [D(self.fview(self.g_a), range(64), R)], H, W)
            cur = self._resblocks(sg, 'Decoder_res_2', self.hp.num_ResB_Dec, self.g_a, self.g_t, self.g_b, H, W, 1)
            self.conv(sg, 'Dec_last1_2', [self.fsrc(cur, 0)], [D(self.fview(self.g_t), range(64), R)], H, W)
            fin = self.finals[it]
            self.conv(sg, 'Dec_last2_2', [self.fsrc(self.g_t, 0)],
                      [D(self.tview(fin[0]), range(0, 3), res=self.tview(self.sharp1, 0)),
                       D(self.tview(fin[1]), range(3, 6), res=self.tview(self.sharp1, 3)),
                       D(self.tview(fin[2]), range(6, 9), res=self.tview(self.stnew))], H, W)

    # ------------------------------------------------------------------------------------------------
    # execution
    # ------------------------------------------------------------------------------------------------
    def _run(self, ops, stream):
        lib, H, W = self.lib, self.H, self.W
        hw4 = H * W * 4
        for op in ops:
            k = op[0]
            if k == 'conv':
                i = op[1]
                self.launch_conv(i, stream, op[2])
            elif k == 'pack':
                _, arr, dst, nch = op
                L.check(lib.demfi_pack_planes(arr, nch, dst.data_ptr(), self.dt, nch, H, W, stream), k)
            elif k == 's2d':
                L.check(lib.demfi_space_to_depth(self.x.data_ptr(), self.s2d.data_ptr(), self.dt, H, W, stream), k)
            elif k == 'overlay':
                L.check(lib.demfi_overlay_mean(self.x.data_ptr(), self.overlay.data_ptr(), H, W, stream), k)
            elif k == 'fgac':
                b = op[1]
                src, dst = self.fview(self.rk, b=b), self.fview(self.smp, b=b)
                flow = self.ffo.data_ptr() + (0 if b == 0 else 2) * hw4
                L.check(lib.demfi_fgac_gather(C.byref(src), flow, C.byref(dst), 64, H, W, None, stream), k)
            elif k == 'gate':
                b = op[1]
                s, e, o = self.fview(self.enc, b=b), self.fview(self.E, b=b), self.fview(self.aF, b=b)
                L.check(lib.demfi_gate_blend(self.gate.data_ptr() + b * hw4, C.byref(s), C.byref(e), C.byref(o), 64, H, W,
                                             stream), k)
            elif k == 'cfr':
                L.check(lib.demfi_cfr_flow_align(self.ffo.data_ptr(), self.ffo.data_ptr() + 2 * hw4, self.t_dev.data_ptr(),
                                                 H, W, self.cfr_acc.data_ptr(), self.ft.data_ptr(), None, stream), k)
            elif k == 'warp_fat':
                _, buf, ba, bb, flows, lbuf, lch, obuf, ob, occ_i = op
                A, B = self.fview(buf, b=ba), self.fview(buf, b=bb)
                O = self.fview(obuf, b=ob or 0)
                fp = flows.data_ptr()
                occ_out = None if occ_i is None else self.occ.data_ptr() + occ_i * hw4
                L.check(lib.demfi_warp_blend(C.byref(A), fp, C.byref(B), fp + 2 * hw4, lbuf.data_ptr() + lch * hw4,
                                             self.t_dev.data_ptr(), C.byref(O), 64, H, W, occ_out, None, stream), k)
            elif k == 'warp_thin':
                it = op[1]
                dn = self.delta[it + 1]
                A, B, O = self.tview(self.sharp1, 0), self.tview(self.sharp1, 3), self.tview(self.stnew)
                fp = dn.data_ptr()
                L.check(lib.demfi_warp_blend(C.byref(A), fp, C.byref(B), fp + 2 * hw4, fp + 4 * hw4, self.t_dev.data_ptr(),
                                             C.byref(O), 3, H, W, self.occ.data_ptr() + (it + 1) * hw4, None, stream), k)
            else:
                raise AssertionError(k)

    def run_trunk(self, stream):
        self._run(self.seg_trunk, stream)

    def run_t(self, stream, n_updates):
        if not 1 <= n_updates <= self.N:
            raise ValueError('num_update=%d outside 1..%d the engine was built for' % (n_updates, self.N))
        self._run(self.seg_t_head, stream)
        for it in range(n_updates):
            self._run(self.seg_iter[it], stream)

    def profile(self, n_updates, reps=3):
        """Per-launch durations (ms, best of ``reps``) of the trunk and one per-t pass, measured with HIP events on
        the stream the kernels are launched on.  Returns a list of (segment, op kind, name, ms, macs)."""
        stream = torch.cuda.current_stream(self.device)
        h = stream.cuda_stream
        segs = [('trunk', self.seg_trunk), ('t_head', self.seg_t_head)] + \
               [('iter%d' % i, self.seg_iter[i]) for i in range(n_updates)]
        out = []
        for sname, ops in segs:
            for op in ops:
                best = 1e30
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    self._run([op], h)
                    e1.record(stream)
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                name = op[2] if op[0] == 'conv' else op[0]
                macs = 0
                if op[0] == 'conv':
                    macs = self.macs[op[2] + '#%d' % (op[1] + 1)]
                out.append((sname, op[0], name, best, macs))
        return out

    def n_launches(self, n_updates):
        return len(self.seg_trunk), len(self.seg_t_head) + sum(len(s) for s in self.seg_iter[:n_updates])
