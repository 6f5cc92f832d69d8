"""Host binding of the forward context of ``libdemfi_hip.so`` (include/demfi_hip.h, ``demfi_ctx_*``).

The launch plan of the DeMFI-Net_rb forward -- buffer set, weight repack, one ``demfi_conv`` descriptor per convolution
call site, launch order -- lives in the C++ library (``demfi_amd/csrc/ctx.cpp``) behind the C ABI, so that any host can
run a forward (``tests/c/forward_golden.c`` does it from plain C).  This module only
  * hands the state_dict to ``demfi_load_weight`` and ONE zero-filled torch allocation to ``demfi_ctx_bind``
    (PyTorch is used for device memory and the stream only; every arithmetic op runs in the HIP library, no fallback),
  * exposes the named buffers of the contexts as tensor views of that allocation (module outputs, runner I/O),
  * forwards ``run_trunk`` / ``run_t`` to ``demfi_forward_trunk`` / ``demfi_forward_t``.

``Plan`` is the descriptor-level companion used by the kernel tests and probes: single convolutions built with
``demfi_conv_build`` (the same C++ routine the context uses) on caller-made buffers.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .spec import HyperParams


def _np_f32(t):
    return np.ascontiguousarray(t.detach().to('cpu', torch.float32).numpy())


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
    """Single-convolution plans for kernel-level tests / probes: owns its buffers, a packed weight blob, a descriptor array
    and a list of launch ops.  The descriptor layout logic is the library's (``demfi_conv_build``)."""

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
        self.macs = {}            # name -> MACs per launch (algorithmic, unpadded)

    def _fat(self, h, w, c, batch=1):
        t = torch.zeros((batch, h, w, c), dtype=self.dtype, device=self.device)
        self._keep.append(t)
        return t

    def _thin(self, c, h=None, w=None):
        t = torch.zeros((c, h or self.H, w or self.W), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def regions(self):
        """Tensors a raw pointer of this plan may point into (tests/plan_sim.py)."""
        return list(self._keep) + ([self.weight_blob] if hasattr(self, 'weight_blob') else [])

    def fsrc(self, buf, cin0, c0=0, nch=None, b=None, up=0):
        """Input piece from a fat buffer [B,h,w,C]: channels [c0,c0+nch) feed original cin [cin0, cin0+nch).
        b=None keeps the batch stride (batched conv), b=k pins image k."""
        B, h, w, Ct = buf.shape
        nch = Ct - c0 if nch is None else nch
        ptr = buf.data_ptr() + (c0 + (0 if b is None else b * h * w * Ct)) * self.esz
        return _Src(True, ptr, Ct, w * Ct, 1, h * w * Ct if b is None else 0, self.f32, range(cin0, cin0 + nch), up)

    def fsrc_map(self, buf, cin, b=0):
        """Input piece = ALL channels of a fat buffer with an explicit channel -> original-cin list (-1 = unused
        padding channel, gets zero weights).  b=k pins image k, b=None keeps the batch stride (batched conv)."""
        B, h, w, Ct = buf.shape
        assert len(cin) == Ct
        ptr = buf.data_ptr() + (0 if b is None else b) * h * w * Ct * self.esz
        return _Src(True, ptr, Ct, w * Ct, 1, h * w * Ct if b is None else 0, self.f32, cin, 0)

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

    def conv(self, seg, name, srcs, dsts, H, W, stride=1, batch=1, weight=None, bias=None, pack=None):
        """Append one convolution launch to segment list ``seg``.  H, W: OUTPUT size.
        pack: (NHWC view, [record channel of octet 0, 1, ...]) -- packed fp16 copy of a thin layer's outputs (demfi_conv.pack)."""
        if weight is None:
            weight = self.sd[name + '.weight']
            bias = self.sd[name + '.bias']
        if weight.dim() == 5:
            weight = weight[:, :, 0]
        wnp = _np_f32(weight)
        bnp = _np_f32(bias)
        cout, cin, kh, kw = wnp.shape
        keep = []
        cs = (L.ConvSrc * len(srcs))()
        for i, s in enumerate(srcs):
            arr = np.asarray(s.cin, np.int32)
            keep.append(arr)
            cs[i] = L.ConvSrc(_view(s.ptr, s.sx, s.sy, s.sc, s.sb, s.is_f32), 1 if s.fat else 0, s.up, len(s.cin), 0,
                              arr.ctypes.data_as(C.POINTER(C.c_int32)))
        cd = (L.ConvDst * len(dsts))()
        for i, d in enumerate(dsts):
            arr = np.asarray(d.couts, np.int32)
            keep.append(arr)
            cd[i] = L.ConvDst(d.view, d.res or _NULL_VIEW, d.aux or _NULL_VIEW, d.act, d.mode, d.scale, d.dy, d.dx, len(d.couts),
                              arr.ctypes.data_as(C.POINTER(C.c_int32)))
        desc = L.Conv()
        nbytes, cout_pad = C.c_int64(0), C.c_int32(0)
        args = (self.dt, H, W, stride, batch, wnp.ctypes.data, bnp.ctypes.data, cout, cin, kh, kw, cs, len(srcs), cd, len(dsts),
                C.byref(desc))
        L.check(self.lib.demfi_conv_build(*args, None, C.byref(nbytes), None, C.byref(cout_pad)), 'conv_build ' + name)
        packed = np.empty(nbytes.value, np.uint8)
        bpk = np.zeros(cout_pad.value, np.float32)
        L.check(self.lib.demfi_conv_build(*args, packed.ctypes.data, C.byref(nbytes), bpk.ctypes.data, C.byref(cout_pad)),
                'conv_build ' + name)
        for g in range(4):
            desc.pack_oct_ch[g] = -1
        if pack is not None:
            desc.pack = pack[0]
            for g, chn in enumerate(pack[1]):
                desc.pack_oct_ch[g] = chn
        desc.wpack = self._add_blob(packed)                 # offsets for now, rebased in _upload()
        desc.bias = self._add_blob(bpk.view(np.uint8))
        self._descs.append(desc)
        self.macs[name + '#%d' % len(self._descs)] = cout * cin * kh * kw * H * W * batch
        seg.append(('conv', len(self._descs) - 1, name))

    def _add_blob(self, arr_u8):
        off = self._wbytes
        self._wblobs.append((off, arr_u8))
        self._wbytes = (off + arr_u8.nbytes + 255) & ~255
        return off

    def _upload(self):
        host = np.zeros(max(self._wbytes, 256), np.uint8)
        for off, a in self._wblobs:
            host[off:off + a.nbytes] = a.reshape(-1)
        self.weight_blob = torch.from_numpy(host).to(self.device)
        self._wblobs = None
        base = self.weight_blob.data_ptr()
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._keep.append(self.zero_page)
        for d in self._descs:
            d.wpack = base + (d.wpack or 0)
            d.bias = base + (d.bias or 0)
            d.zero_page = self.zero_page.data_ptr()
        sz = C.sizeof(L.Conv)
        raw = bytearray(len(self._descs) * sz)
        for i, d in enumerate(self._descs):
            raw[i * sz:(i + 1) * sz] = bytes(d)
        self.desc_dev = torch.frombuffer(raw, dtype=torch.uint8).clone().to(self.device)
        self._desc_sz = sz

    def launch_conv(self, i, stream, what='conv'):
        L.check(self.lib.demfi_conv2d(C.byref(self._descs[i]), self.desc_dev.data_ptr() + i * self._desc_sz, stream), what)

    def launch_resblock(self, i1, i2, stream, what='resblock'):
        """Descriptors i1 (conv1 -> ReLU) and i2 (conv2 + identity) as ONE launch of the fused residual-block kernel."""
        L.check(self.lib.demfi_resblock3x3_c64(C.byref(self._descs[i1]), C.byref(self._descs[i2]), stream), what)


    def launch_gru_r(self, i, stream, what='gru_r'):
        """Descriptor i (the reset-gate layer with the MUL epilogue) on the round-6 SepConvGRU kernel (demfi_gru_r)."""
        L.check(self.lib.demfi_gru_r(C.byref(self._descs[i]), stream), what)

    def launch_gru_zq(self, iz, iq, stream, what='gru_zq'):
        """Descriptors iz (update gate, sigmoid -> z buffer) and iq (candidate, GRU epilogue) as ONE launch: z stays on chip."""
        L.check(self.lib.demfi_gru_zq(C.byref(self._descs[iz]), C.byref(self._descs[iq]), stream), what)


SEG_TRUNK, SEG_HEAD, SEG_ITER, SEG_TB_HEAD, SEG_TB_ITER = 0, 1, 2, 3, 4
KIND_NAME = {0: 'conv', 1: 'pack', 2: 's2d', 3: 'overlay', 4: 'fgac', 5: 'gate', 6: 'cfr', 7: 'warp', 8: 'fgac_window', 9: 'avg_pool',
             10: 'resblock',        # resblock: ONE launch for conv1 -> ReLU -> conv2 + identity (op.conv / op.nch = the two descriptors)
             11: 'gru_r', 12: 'gru_zq', 13: 'viz'}   # round 6: SepConvGRU half-step as r*h, then z + q + blend in one launch (op.conv / op.nch = convz / convq)


class Engine:
    """One forward context (frame size, dtype, max recursion depth, n_trunk x n_ctx buffer sets) of the HIP library.

    ``n_ctx`` > 1 builds several independent per-t buffer sets so that different time instants t of one window can run
    concurrently on different streams (they only share the read-only trunk outputs); ``n_trunk`` > 1 builds several trunk
    buffer sets, each with its own per-t sets, so that the trunk of the next window can run while the time instants of
    the current one are in flight (WindowRunner).  ``use_ctx(c, trunk)`` selects which set the convenience attributes
    (``x``, ``t_dev``, ``sharp1``, ``finals``, ``delta``, ``occ``, ``overlay``) refer to."""

    def __init__(self, state_dict, H, W, dtype=torch.float16, device='cuda:0', max_updates=3, hp=None, n_ctx=1, n_trunk=1):
        if H % 8 or W % 8:
            raise ValueError('DeMFI-Net needs H, W multiples of 8 (the harness pads to 32): got %dx%d' % (H, W))
        self.lib = L.load()
        self.H, self.W, self.N = H, W, max_updates
        self.dtype = dtype
        self.f32 = dtype == torch.float32
        self.dt = L.F32 if self.f32 else L.F16
        self.device = torch.device(device)
        self.hp = hp or HyperParams()
        if self.hp.nf != 64 or self.hp.scale_factor != 2:
            raise NotImplementedError('the HIP path is built for nf=64, scale_factor=2 (the released configuration)')
        chp = L.HParams(self.hp.nf, self.hp.scale_factor, self.hp.num_ResB_FACFB, self.hp.num_ResB_Dec,
                        1 if self.hp.shared_FGAC_flag else 0, getattr(self.hp, 'fgac_rr', 0), getattr(self.hp, 'fgac_sr', 0),
                        (1 if getattr(self.hp, 'fgac_map', 0) else 0) | (2 if getattr(self.hp, 'extras', False) else 0))
        self.extras = bool(getattr(self.hp, 'extras', False))
        self._ctx = C.c_void_p()
        self._n_trunk, self._n_ctx = max(1, n_trunk), max(1, n_ctx)
        L.check(self.lib.demfi_ctx_create(H, W, max_updates, self.dt, C.byref(chp), self._n_trunk, self._n_ctx,
                                          C.byref(self._ctx)), 'ctx_create')
        for key, val in state_dict.items():
            a = _np_f32(val)
            shp = (C.c_int64 * a.ndim)(*a.shape)
            L.check(self.lib.demfi_load_weight(self._ctx, key.encode(), a.ctypes.data, shp, a.ndim), 'load_weight ' + key)
        nbytes = self.lib.demfi_ctx_workspace_bytes(self._ctx)
        on_host = self.device.type != 'cuda'
        # ONE allocation: packed weights | descriptors | every activation buffer of every context (zero-filled: the CFR
        # accumulators rely on it)
        self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._ws_off = (-self.workspace.data_ptr()) % 256
        self._base = self.workspace.data_ptr() + self._ws_off
        stream = 0 if on_host else torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.demfi_ctx_bind(self._ctx, self._base, nbytes, 1 if on_host else 0, stream), 'ctx_bind')
        off, nb = C.c_int64(0), C.c_int64(0)
        L.check(self.lib.demfi_ctx_weight_region(self._ctx, C.byref(off), C.byref(nb)))
        self.weight_blob = self.workspace[self._ws_off + off.value:self._ws_off + off.value + nb.value]
        self._views = {}
        self._ctxs = [[self._ctx_dict(k, c) for c in range(self._n_ctx)] for k in range(self._n_trunk)]
        self._trunks = [self._trunk_dict(k) for k in range(self._n_trunk)]
        self._tb = [self._tb_dict(k) for k in range(self._n_trunk)] if self._n_ctx > 1 else None
        self.use_ctx(0, trunk=0)

    def __del__(self):
        try:
            if getattr(self, '_ctx', None):
                self.lib.demfi_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- named buffers -------------------------------------------------------------------------------------
    def buffer(self, name, trunk=0, c=-1):
        """Tensor view of a named plan buffer (c = -1: trunk buffers).  fat: [B,h,w,C] path dtype; thin: [C,h,w] fp32."""
        key = (name, trunk, c)
        t = self._views.get(key)
        if t is None:
            off, kind, dims = C.c_int64(0), C.c_int32(0), (C.c_int32 * 4)()
            L.check(self.lib.demfi_ctx_buffer(self._ctx, trunk, c, name.encode(), C.byref(off), C.byref(kind), dims), 'ctx_buffer')
            d = list(dims)
            lo = self._ws_off + off.value
            if kind.value == 0:
                n = d[0] * d[1] * d[2] * d[3] * (4 if self.f32 else 2)
                t = self.workspace[lo:lo + n].view(self.dtype).view(d)
            elif kind.value == 1:
                n = d[0] * d[1] * d[2] * 4
                t = self.workspace[lo:lo + n].view(torch.float32).view(d[:3])
            else:
                t = self.workspace[lo:lo + d[0] * 8].view(torch.int64)
            self._views[key] = t
        return t

    def _trunk_dict(self, k):
        H, W = self.H, self.W
        d = {'x': self.buffer('x', k).view(3, 4, H, W), 'overlay': self.buffer('overlay', k), 'ffo': self.buffer('ffo', k),
             'aF': self.buffer('aF', k), 'F01': self.buffer('F01', k), 'gate': self.buffer('gate', k)}
        if self.extras:                                       # maps of the visualisation / training return tuples (DEMFI_HP_EXTRAS)
            d['viz'] = self.buffer('viz', k).view(2, 6, H, W)
        return d

    def _ctx_dict(self, k, c):
        H, W, N = self.H, self.W, self.N
        return {'t_dev': self.buffer('t', k, c).view(1), 'sharp1': self.buffer('sharp1', k, c),
                'finals': self.buffer('finals', k, c).view(N, 3, 3, H, W), 'delta': self.buffer('delta', k, c).view(N + 1, 5, H, W),
                'occ': self.buffer('occ', k, c), 'ft': self.buffer('ft', k, c), 'cfr_acc': self.buffer('cfr_acc', k, c),
                'sink': self.buffer('sink', k, c)}

    def _tb_dict(self, k):
        """The per-t contexts' "t" and "sink" buffers of trunk set k as ONE tensor each (the copies of a per-t buffer are
        contiguous in the workspace): t_col [n_ctx] fp32 (strided view), sink_all [n_ctx, 32] int64."""
        n = self._n_ctx
        t0, t1 = self.buffer('t', k, 0), self.buffer('t', k, 1)
        s0, s1 = self.buffer('sink', k, 0), self.buffer('sink', k, 1)
        t_stride, s_stride = t1.data_ptr() - t0.data_ptr(), s1.data_ptr() - s0.data_ptr()
        if t_stride % 4 or s_stride != 256:
            raise RuntimeError('unexpected per-t context strides (%d, %d)' % (t_stride, s_stride))
        lo = t0.data_ptr() - self.workspace.data_ptr()
        t_col = self.workspace[lo:lo + n * t_stride].view(torch.float32).view(n, t_stride // 4)[:, 0]
        lo = s0.data_ptr() - self.workspace.data_ptr()
        sink_all = self.workspace[lo:lo + n * 256].view(torch.int64).view(n, 32)
        return {'t_col': t_col, 'sink_all': sink_all}

    @property
    def n_ctx(self):
        return self._n_ctx

    @property
    def n_trunk(self):
        return self._n_trunk

    def use_ctx(self, c, trunk=None):
        """Bind trunk context ``trunk`` (default: the current one) and its per-t context c to this engine's attributes."""
        if trunk is not None or not hasattr(self, 'trunk'):
            self.trunk = trunk or 0
            self.__dict__.update(self._trunks[self.trunk])
        self.__dict__.update(self._ctxs[self.trunk][c])
        self.ctx = c

    def reset_cfr(self, stream=None):
        """Re-zero the splat accumulators of every per-t context (they clean themselves after every completed launch; call
        this after an aborted one: a failed graph replay leaves garbage that every later call would add)."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        for row in self._ctxs:
            for ctx in row:
                L.check(self.lib.demfi_cfr_reset(ctx['cfr_acc'].data_ptr(), self.H, self.W, st), 'cfr_reset')

    def regions(self):
        return [self.workspace]

    def activation_bytes(self):
        return int(self.workspace.numel()) - int(self.weight_blob.numel())

    # ---- execution --------------------------------------------------------------------------------------------
    def run_trunk(self, stream):
        L.check(self.lib.demfi_forward_trunk(self._ctx, self.trunk, None, stream), 'forward_trunk')

    def ingest_u8(self, frame_ptrs, h, w, stream):
        """4 BGR uint8 [h,w,3] device frames (ctypes array of 4 pointers) -> x, s2d, overlay of the bound trunk context."""
        L.check(self.lib.demfi_ingest_u8(self._ctx, self.trunk, frame_ptrs, h, w, stream), 'ingest_u8')

    def run_trunk_body(self, stream):
        """The trunk without its s2d / overlay prologue (what follows ingest_u8)."""
        L.check(self.lib.demfi_forward_trunk_body(self._ctx, self.trunk, stream), 'forward_trunk_body')

    @property
    def supports_u8_sink(self):
        """The uint8 egress is an epilogue of the fp16 thin-output kernel."""
        return not self.f32

    def run_t(self, stream, n_updates):
        if not 1 <= n_updates <= self.N:
            raise ValueError('num_update=%d outside 1..%d the engine was built for' % (n_updates, self.N))
        L.check(self.lib.demfi_forward_t(self._ctx, self.trunk, self.ctx, n_updates, stream), 'forward_t')

    def run_tb(self, stream, n_updates, final_only=False):
        """The per-t segment of ALL per-t contexts of the bound trunk set as one launch sequence (convolutions batched over
        the contexts); every context reads its own t / sink buffers.  final_only: the warp + D2 tail of the recursions before
        the last one is not run (their frames are outputs only: nothing later reads them)."""
        if not 1 <= n_updates <= self.N:
            raise ValueError('num_update=%d outside 1..%d the engine was built for' % (n_updates, self.N))
        f = self.lib.demfi_forward_tb_final if final_only else self.lib.demfi_forward_tb
        L.check(f(self._ctx, self.trunk, n_updates, stream), 'forward_tb')

    # ---- introspection -------------------------------------------------------------------------------------------
    def ops(self, segment, it=0, trunk=None, c=None):
        """The launch ops (L.Op structs) of one segment of one context."""
        trunk = self.trunk if trunk is None else trunk
        c = self.ctx if c is None else c
        n = L.check(self.lib.demfi_ctx_num_ops(self._ctx, segment, trunk, c, it), 'num_ops')
        out = []
        for i in range(n):
            op = L.Op()
            L.check(self.lib.demfi_ctx_get_op(self._ctx, segment, trunk, c, it, i, C.byref(op)), 'get_op')
            out.append(op)
        return out

    def conv_desc(self, index):
        p = self.lib.demfi_ctx_conv_desc(self._ctx, index)
        if not p:
            raise IndexError(index)
        return p.contents

    @property
    def n_convs(self):
        return self.lib.demfi_ctx_num_convs(self._ctx)

    def op_algorithmic_bytes(self, op):
        """ALGORITHMIC HBM bytes of one launch: every tensor it reads or writes counted once -- input pieces (an
        image a batched launch re-reads with batch stride 0 counts once), residual / aux, outputs, packed weights; planar operands at 4 B.
        What bench.py prices against the 8 TB/s ceiling in its per-op table; the measured bytes are the PMC figures beside it."""
        H, W = self.H, self.W
        kind = KIND_NAME.get(op.kind, '')
        esz = 4 if self.f32 else 2
        nb = max(1, int(op.bt.nb))

        def conv_in(d):
            tot = 0
            for i in range(d.n_pieces):
                pc = d.pieces[i]
                if not pc.v.ptr:
                    continue
                h_in, w_in = (d.inH >> pc.up_shift, d.inW >> pc.up_shift) if pc.up_shift else (d.inH, d.inW)
                tot += pc.nch * (4 if pc.v.is_f32 else 2) * h_in * w_in * (d.batch if pc.v.sb else 1)
            return tot

        def conv_out(d, with_aux=True):
            tot = 0
            for sg in range(d.n_segs):
                g = d.segs[sg]
                n = sum(d.oct_n[o] for o in range(d.cout_pad // 8) if d.oct_seg[o] == sg)
                px = d.H * d.W * d.batch * (g.scale * g.scale if g.scale > 1 else 1) // (g.scale * g.scale if g.scale > 1 else 1)
                tot += n * (4 if g.dst.is_f32 else 2) * px
                if g.res.ptr:
                    tot += n * (4 if g.res.is_f32 else 2) * d.H * d.W * (d.batch if g.res.sb else 1)
                if with_aux and g.aux.ptr:
                    tot += n * (4 if g.aux.is_f32 else 2) * d.H * d.W * (d.batch if g.aux.sb else 1)
            if d.pack.ptr:
                tot += 16 * 2 * d.H * d.W * d.batch
            return tot

        def weights(d):
            return sum(d.chunks[c].nks for c in range(d.n_chunks)) * 32 * d.kh * d.kw * d.cout_pad
        if kind == 'conv':
            d = self.conv_desc(op.conv)
            return conv_in(d) + conv_out(d) + weights(d)
        if kind == 'gru_r':                                       # h, x in; r*h out (the residual IS the h window)
            d = self.conv_desc(op.conv)
            return conv_in(d) + 64 * esz * d.H * d.W * d.batch + weights(d)
        if kind == 'resblock':                                    # input + output once; the intermediate stays in LDS
            d1, d2 = self.conv_desc(op.conv), self.conv_desc(op.nch)
            return conv_in(d1) + 64 * esz * d2.H * d2.W * d2.batch + weights(d1) + weights(d2)
        if kind == 'gru_zq':                                      # h, x, r*h in; h' out; z stays on chip
            dz, dq = self.conv_desc(op.conv), self.conv_desc(op.nch)
            return conv_in(dz) + 64 * esz * dq.H * dq.W * dq.batch * 2 + weights(dz) + weights(dq)
        px = H * W * nb
        if kind == 'warp':
            C_ = op.nch
            e = esz if C_ == 64 else 4
            shared = C_ == 64 and nb > 1 and not op.bt.a and not op.bt.b     # trunk features shared by the time instants of a window
            return (2 * C_ * e * H * W * (1 if shared else nb)) + (C_ * e + 20) * px + (8 * esz * px if op.p[4] else 0)
        if kind == 'cfr':
            return 32 * px
        if kind == 'pack':
            return (sum(1 for i in range(32) if op.p[i]) * 4 + op.nch * esz) * px
        if kind in ('fgac', 'gate'):
            return ((2 if kind == 'fgac' else 3) * op.nch * esz + (8 if kind == 'fgac' else 4)) * H * W
        if kind == 's2d':
            return 12 * 4 * H * W + 48 * esz * (H // 2) * (W // 2)
        if kind == 'overlay':
            return 9 * 4 * H * W
        if kind == 'viz':
            return (op.nch * esz * (2 if op.b.ptr else 1) + 4) * H * W if op.conv == 0 else 8 * H * W
        return 0

    def run_op(self, op, stream):
        L.check(self.lib.demfi_run_op(self._ctx, C.byref(op), stream), 'run_op')

    def profile(self, n_updates, reps=5, isolated=False, batched=False):
        """Per-launch durations (ms, MEAN over ``reps`` passes after one warm-up pass) of the trunk and one per-t pass, measured
        with HIP events on the stream the kernels are launched on.  Default: IN SEQUENCE -- the whole forward is launched op
        after op with an event between consecutive launches, so every kernel sees the cache state the real pipeline leaves it
        (a kernel timed in a loop of its own re-reads inputs that the 256 MB Infinity Cache kept from the previous repetition:
        warp_blend looked 25 % faster that way).  ``isolated=True`` is that per-op loop (kernel tuning only).
        Returns a list of (segment, op kind, name, ms, macs, contexts in the launch, algorithmic HBM bytes)."""
        stream = torch.cuda.current_stream(self.device)
        h = stream.cuda_stream
        if batched:                                           # the batched per-t plan: every launch covers all n_ctx contexts
            segs = [('trunk', self.ops(SEG_TRUNK)), ('t_head', self.ops(SEG_TB_HEAD))] + \
                   [('iter%d' % i, self.ops(SEG_TB_ITER, i)) for i in range(n_updates)]
        else:
            segs = [('trunk', self.ops(SEG_TRUNK)), ('t_head', self.ops(SEG_HEAD))] + \
                   [('iter%d' % i, self.ops(SEG_ITER, i)) for i in range(n_updates)]
        flat = [(sname, op) for sname, ops in segs for op in ops]
        tot = [0.0] * len(flat)
        if isolated:
            # under the workspace arena an op replayed out of plan order reads memory later tenants have recycled (wrong access
            # patterns, possibly NaNs): the isolated loop is only meaningful on the unaliased layout
            import os
            if os.environ.get('DEMFI_ARENA', '1') != '0':
                raise RuntimeError('Engine.profile(isolated=True) replays single ops out of plan order: run it with DEMFI_ARENA=0 '
                                   '(one memory region per buffer); the default in-sequence profile needs nothing')
            for i, (_, op) in enumerate(flat):
                self.run_op(op, h)
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    self.run_op(op, h)
                    e1.record(stream)
                    e1.synchronize()
                    tot[i] += e0.elapsed_time(e1)
        else:
            for rep in range(reps + 1):
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(flat) + 1)]
                evs[0].record(stream)
                for i, (_, op) in enumerate(flat):
                    self.run_op(op, h)
                    evs[i + 1].record(stream)
                evs[-1].synchronize()
                if rep:                                           # pass 0 warms up
                    for i in range(len(flat)):
                        tot[i] += evs[i].elapsed_time(evs[i + 1])
        out = []
        for (sname, op), t in zip(flat, tot):
            kind = KIND_NAME.get(op.kind, str(op.kind))
            if kind == 'warp':
                kind = 'warp_fat' if op.nch == 64 else 'warp_thin'
            out.append((sname, kind, op.name.decode(), t / reps, int(op.macs), max(1, int(op.bt.nb)),    # [5]: per-t contexts covered by a batched point-wise launch
                        int(self.op_algorithmic_bytes(op))))                                            # [6]: algorithmic HBM bytes of the launch
        return out

    def n_launches(self, n_updates):
        lib, cx = self.lib, self._ctx
        return (lib.demfi_ctx_num_ops(cx, SEG_TRUNK, 0, 0, 0),
                lib.demfi_ctx_num_ops(cx, SEG_HEAD, 0, 0, 0) + sum(lib.demfi_ctx_num_ops(cx, SEG_ITER, 0, 0, i)
                                                                   for i in range(n_updates)))
