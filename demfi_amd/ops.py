"""Single-call operators of the hot path: ``SepConvGRU`` and ``FGAC`` on the HIP kernels, outside a whole-network context.

SURVEY.md section 8(b) names two operator-level entry points besides the segment-level ones (``demfi_gru_sep``, ``demfi_fgac``).  In
the library they exist as kernel-level calls (``demfi_conv2d`` with the GRU epilogues, ``demfi_fgac_gather``, ``demfi_gate_blend``) that
the network plan strings together; this module strings the same calls together for ONE operator and gives it the reference module's
surface -- constructor arguments, ``state_dict`` keys, ``forward`` arguments and return values:

  * ``SepConvGRU(h_dim, x_dim)``, ``forward(h, x) -> h``                        /root/reference/DeMFInet.py:827-857
  * ``FGAC(args)``, ``forward(ref, source, flow_s2r) -> (out, w_sr, diff)``     /root/reference/DeMFInet.py:361-496 (rr = sr = 0)

Tensors are ordinary NCHW torch tensors on the GPU; PyTorch converts the layout (NCHW <-> the kernels' NHWC records) and owns the
memory, every arithmetic op of the operator runs in ``libdemfi_hip.so`` (no fallback: without the library ``_lib.load()`` raises).
The launch plan of a (batch, H, W) shape is built once and cached.
"""
import ctypes as C

import torch

from . import _lib as L
from .engine import Plan, _Dst


def _strip(sd, prefix):
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v.detach().to('cpu', torch.float32).contiguous()
    return out


class _Operator:
    _keys = ()

    def __init__(self, dtype=torch.float16, device='cuda:0'):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError('dtype must be torch.float16 or torch.float32')
        self.dtype, self.device = dtype, torch.device(device)
        self.sd = None
        self._plans = {}

    def load_state_dict(self, state_dict, prefix=''):
        """Takes the reference module's keys (``<prefix>convz1.weight`` ...); missing keys raise like ``nn.Module.load_state_dict``."""
        sd = _strip(state_dict, prefix)
        missing = [k for k in self._keys if k not in sd]
        if missing:
            raise KeyError('missing keys in state_dict: %s' % ', '.join(missing))
        self.sd = sd
        self._plans = {}
        return self

    def _check(self, name, t, ch):
        if self.sd is None:
            raise RuntimeError('load_state_dict() first')
        if t.dim() != 4 or t.shape[1] != ch:
            raise ValueError('%s must be [B,%d,H,W], got %s' % (name, ch, tuple(t.shape)))
        if t.device != self.device:
            raise ValueError('%s is on %s, the operator on %s' % (name, t.device, self.device))

    @staticmethod
    def _to_nhwc(buf, t):
        buf.copy_(t.permute(0, 2, 3, 1))

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    __call__ = lambda self, *a: self.forward(*a)


class SepConvGRU(_Operator):
    """Separable convolutional GRU (DeMFInet.py:827-857): a horizontal (1x5) and a vertical (5x1) GRU step.  Per step two launches:
    the fused z | r convolution (sigmoid; sigmoid * h) and the q convolution with the GRU blend ``(1 - z) h + z tanh(.)`` in its
    epilogue -- for 64 + 64 channels in fp16 the persistent 1x5 / 5x1 kernel, otherwise the general kernel."""
    _keys = tuple('conv%s%d.%s' % (g, i, p) for i in (1, 2) for g in 'zrq' for p in ('weight', 'bias'))

    def __init__(self, h_dim=64, x_dim=64, dtype=torch.float16, device='cuda:0'):
        super().__init__(dtype, device)
        self.h_dim, self.x_dim = h_dim, x_dim

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        hd, xd = self.h_dim, self.x_dim
        pl = Plan(H, W, self.dtype, self.device)
        bufs = {n: pl._fat(H, W, hd, B) for n in ('h0', 'z', 'rh', 'h1', 'h2')}
        bufs['x'] = pl._fat(H, W, xd, B)
        seg = []
        for i, (hin, hout) in ((1, ('h0', 'h1')), (2, ('h1', 'h2'))):
            wzr = torch.cat([self.sd['convz%d.weight' % i], self.sd['convr%d.weight' % i]], 0)
            bzr = torch.cat([self.sd['convz%d.bias' % i], self.sd['convr%d.bias' % i]], 0)
            h = bufs[hin]
            pl.conv(seg, 'convzr%d' % i, [pl.fsrc(h, 0), pl.fsrc(bufs['x'], hd)],
                    [_Dst(pl.fview(bufs['z']), range(0, hd), L.ACT_SIGMOID),
                     _Dst(pl.fview(bufs['rh']), range(hd, 2 * hd), mode=L.MODE_MUL, res=pl.fview(h))],
                    H, W, batch=B, weight=wzr, bias=bzr)
            pl.conv(seg, 'convq%d' % i, [pl.fsrc(bufs['rh'], 0), pl.fsrc(bufs['x'], hd)],
                    [_Dst(pl.fview(bufs[hout]), range(hd), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(bufs['z']))],
                    H, W, batch=B, weight=self.sd['convq%d.weight' % i], bias=self.sd['convq%d.bias' % i])
        pl._upload()
        self._plans[key] = (pl, bufs, len(seg))
        return self._plans[key]

    def forward(self, h, x):
        self._check('h', h, self.h_dim)
        self._check('x', x, self.x_dim)
        B, _, H, W = h.shape
        if x.shape[0] != B or x.shape[2:] != h.shape[2:]:
            raise ValueError('h %s and x %s differ in batch or size' % (tuple(h.shape), tuple(x.shape)))
        pl, bufs, n = self._plan(B, H, W)
        self._to_nhwc(bufs['h0'], h)
        self._to_nhwc(bufs['x'], x)
        st = self._stream()
        for i in range(n):
            pl.launch_conv(i, st, 'SepConvGRU launch %d' % i)
        # an OWNED tensor like the reference nn.Module returns: a view of the cached per-shape plan buffer would be overwritten by the
        # next forward() on the same shape
        return bufs['h2'].permute(0, 3, 1, 2).to(h.dtype, copy=True)


class FGAC(_Operator):
    """Flow-guided attentive correlation at rr = sr = 0 (DeMFInet.py:361-496; the form DeMFI-Net uses): ``conv_ref_k`` ->
    bilinear sample at the ABSOLUTE flow coordinates (zeros outside, align_corners) -> ``fusion`` -> gate ``sigmoid(w_gen_2(relu(w_gen(
    [source, E_s]))))`` -> ``w source + (1 - w) E_s``.  ``conv_source_k`` is accepted and unused (its product is multiplied by a
    softmax over ONE element).  Returns ``(bolstered_F_s, w_sr, diff)`` like the reference with ``visualization_flag`` off; ``diff``,
    the min-max normalised mean |change| the network discards, is computed with torch ops from the kernel's output."""
    _keys = tuple('%s.%s' % (n, p) for n in ('conv_ref_k', 'fusion', 'w_gen', 'w_gen_2') for p in ('weight', 'bias'))

    def __init__(self, args=None, dtype=torch.float16, device='cuda:0'):
        super().__init__(dtype, device)
        self.nf = getattr(args, 'nf', 64) if args is not None else 64

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        nf = self.nf
        pl = Plan(H, W, self.dtype, self.device)
        bufs = {n: pl._fat(H, W, nf, B) for n in ('ref', 'source', 'ref_k', 'sampled', 'e_s', 'hid', 'out')}
        bufs['w'] = pl._thin(B, H, W)                       # one gate plane per image
        sd = self.sd
        seg = []
        pl.conv(seg, 'conv_ref_k', [pl.fsrc(bufs['ref'], 0)], [_Dst(pl.fview(bufs['ref_k']), range(nf))], H, W, batch=B,
                weight=sd['conv_ref_k.weight'], bias=sd['conv_ref_k.bias'])
        pl.conv(seg, 'fusion', [pl.fsrc(bufs['sampled'], 0)], [_Dst(pl.fview(bufs['e_s']), range(nf))], H, W, batch=B,
                weight=sd['fusion.weight'], bias=sd['fusion.bias'])
        pl.conv(seg, 'w_gen', [pl.fsrc(bufs['source'], 0), pl.fsrc(bufs['e_s'], nf)], [_Dst(pl.fview(bufs['hid']), range(nf), L.ACT_RELU)],
                H, W, batch=B, weight=sd['w_gen.weight'], bias=sd['w_gen.bias'])
        pl.conv(seg, 'w_gen_2', [pl.fsrc(bufs['hid'], 0)], [_Dst(pl.tview(bufs['w'], 0, sb=H * W), [0], L.ACT_SIGMOID)], H, W, batch=B,
                weight=sd['w_gen_2.weight'], bias=sd['w_gen_2.bias'])
        pl._upload()
        self._plans[key] = (pl, bufs)
        return self._plans[key]

    def forward(self, ref, source, flow_s2r):
        self._check('ref', ref, self.nf)
        self._check('source', source, self.nf)
        self._check('flow_s2r', flow_s2r, 2)
        B, _, H, W = ref.shape
        if source.shape != ref.shape or flow_s2r.shape[0] != B or flow_s2r.shape[2:] != ref.shape[2:]:
            raise ValueError('ref %s, source %s, flow %s do not match' % (tuple(ref.shape), tuple(source.shape), tuple(flow_s2r.shape)))
        pl, bufs = self._plan(B, H, W)
        lib, st = pl.lib, self._stream()
        self._to_nhwc(bufs['ref'], ref)
        self._to_nhwc(bufs['source'], source)
        flow = flow_s2r.to(torch.float32).contiguous()      # planar [B,2,H,W] fp32, as the kernels read flows
        pl.launch_conv(0, st, 'FGAC conv_ref_k')
        for b in range(B):
            vs, vo = pl.fview(bufs['ref_k'], b=b), pl.fview(bufs['sampled'], b=b)
            L.check(lib.demfi_fgac_gather(C.byref(vs), flow[b].data_ptr(), C.byref(vo), self.nf, H, W, None, st), 'fgac_gather')
        for i in (1, 2, 3):
            pl.launch_conv(i, st, 'FGAC conv %d' % i)
        for b in range(B):
            vs, ve, vo = pl.fview(bufs['source'], b=b), pl.fview(bufs['e_s'], b=b), pl.fview(bufs['out'], b=b)
            L.check(lib.demfi_gate_blend(bufs['w'][b].data_ptr(), C.byref(vs), C.byref(ve), C.byref(vo), self.nf, H, W, st), 'gate_blend')
        out = bufs['out'].permute(0, 3, 1, 2).to(ref.dtype, copy=True)      # owned (not views of the cached plan buffers)
        w_sr = bufs['w'].view(B, 1, H, W).to(ref.dtype, copy=True)
        # visualisation by-product (DeMFInet.py:455-462), not on the network's path: mean |out - source| min-max normalised per image
        diff = (out.float() - source.float()).abs().mean(1, keepdim=True)
        flat = diff.view(B, -1)
        flat = flat - flat.min(1, keepdim=True)[0]
        flat = flat / flat.max(1, keepdim=True)[0]
        return out, w_sr, flat.view(B, 1, H, W)
