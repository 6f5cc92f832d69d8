"""Kernel-level parity on a real MI355X, every call through the C ABI of libdemfi_hip.so.

Comparisons: the oracle (oracle/demfi_oracle.py) and the fixtures frozen from the upstream reference
(tests/golden).  Integer maps (splat target indices, grid-sample floor indices / in-bounds bits / validity)
must be BIT-IDENTICAL; floating-point values are compared with the tolerance written in each test."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import _lib as L                      # noqa: E402
from demfi_amd.engine import Plan, _Dst              # noqa: E402
from oracle import demfi_oracle as O                 # noqa: E402

DEV = 'cuda:0'


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _view_planar(t, c0=0):
    Ct, h, w = t.shape
    return L.View(t.data_ptr() + c0 * h * w * 4, 1, w, h * w, 0, 1, 0)


def _view_nhwc(t):
    h, w, c = t.shape
    return L.View(t.data_ptr(), c, w * c, 1, 0, 1 if t.dtype == torch.float32 else 0, 0)


def test_device_is_gfx950():
    st, name, ncu, mem = L.device_info()
    assert st == 0, name
    assert name.startswith('gfx950') and ncu == 256


# ------------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # cin, cout, kh, kw, stride, H, W (output), act, res
    (64, 64, 3, 3, 1, 16, 64, L.ACT_RELU, True),
    (64, 64, 3, 3, 1, 13, 45, L.ACT_NONE, False),      # ragged tile edges
    (64, 64, 3, 3, 1, 133, 530, L.ACT_RELU, True),     # 289 tiles > 256 workgroups: the persistent tile walk, both accumulator roles
    (64, 32, 3, 3, 1, 141, 499, L.ACT_NONE, False),    # same, one 32-cout subtile
    (48, 96, 5, 5, 1, 16, 32, L.ACT_NONE, False),
    (96, 32, 3, 3, 1, 8, 32, L.ACT_RELU, False),
    (224, 96, 1, 1, 1, 8, 40, L.ACT_NONE, True),
    (192, 64, 7, 7, 1, 16, 32, L.ACT_TANH, False),
    (128, 64, 1, 5, 1, 9, 33, L.ACT_SIGMOID, False),
    (128, 64, 5, 1, 1, 9, 33, L.ACT_NONE, False),
    (64, 128, 4, 4, 2, 8, 32, L.ACT_RELU, False),
    (96, 256, 3, 3, 1, 8, 32, L.ACT_NONE, False),      # two cout blocks
    (64, 133, 3, 3, 1, 8, 32, L.ACT_NONE, False),      # nco = 5, ragged cout
    (32, 5, 3, 3, 1, 8, 32, L.ACT_NONE, False),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_vs_torch(case, dtype):
    cin, cout, kh, kw, stride, H, W, act, with_res = case
    torch.manual_seed(cin * 1000 + cout + kh)
    inH, inW = H * stride, W * stride
    pl = Plan(H, W, dtype, DEV)
    x = pl._fat(inH, inW, cin)
    x.copy_(torch.randn(x.shape, device=DEV))
    out = pl._fat(H, W, cout)
    res = pl._fat(H, W, cout)
    res.copy_(torch.randn(res.shape, device=DEV))
    wt = torch.randn(cout, cin, kh, kw) * (1.0 / (cin * kh * kw) ** 0.5)
    bs = torch.randn(cout) * 0.1
    seg = []
    pl.conv(seg, 'case', [pl.fsrc(x, 0)], [_Dst(pl.fview(out), range(cout), act, res=pl.fview(res) if with_res else None)],
            H, W, stride=stride, weight=wt, bias=bs)
    pl._upload()
    pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    xin = x[0].permute(2, 0, 1).float().cpu()[None]
    wq = wt.half().float() if dtype == torch.float16 else wt
    pad = (1, 1) if stride == 2 else (kh // 2, kw // 2)
    ref = torch.nn.functional.conv2d(xin.double(), wq.double(), bs.double(), stride=stride, padding=pad)[0]
    if with_res:
        ref = ref + res[0].permute(2, 0, 1).double().cpu()
    ref = {L.ACT_NONE: lambda z: z, L.ACT_RELU: torch.relu, L.ACT_TANH: torch.tanh, L.ACT_SIGMOID: torch.sigmoid}[act](ref)
    got = out[0].permute(2, 0, 1).double().cpu()
    # fp32 path: exact-fp32 MFMA, only summation order differs from the fp64 reference;
    # fp16 path: inputs/weights identical fp16 values, fp32 accumulate, fp16 rounding of the stored result.
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    err = (got - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), (case, err)


RESBLOCK_CASES = [
    # H, W, batch: strips are 30 output columns wide, steps 16 rows high
    (16, 30, 1),        # one item: opening step + one step
    (16, 32, 1),        # a second strip with 2 useful columns
    (37, 75, 2),        # ragged last strip / last step, two images
    (48, 64, 3),        # three full steps per strip: the carried lines across steps
    (133, 530, 1),      # 18 strips x 9 steps: several items per workgroup, chains broken at workgroup boundaries
    (200, 1280, 2),     # 43 strips x 13 steps x 2 = 1118 items on 256 workgroups (XCD bands, chains that start mid-strip)
]


@pytest.mark.parametrize('case', RESBLOCK_CASES)
def test_fused_resblock_vs_two_launches_and_torch(case):
    """Round 5: y = x + conv2(relu(conv1(x))) (ResBlock2D / ResBlock2D_3D, DeMFInet.py:524-563) in ONE launch with the intermediate in
    LDS, against (i) the two demfi_conv2d launches it replaces -- conv1's half is bit-identical (same MFMA order, same fp16 rounding
    of the intermediate), conv2 accumulates onto bias + identity instead of adding the identity last, i.e. at most one fp16 ulp
    apart -- and (ii) an fp64 torch reference of the block that rounds the intermediate to fp16 like both forms do.  The scratch
    buffer of the two-launch form must stay untouched by the fused launch."""
    H, W, B = case
    torch.manual_seed(H * 7 + W)
    pl = Plan(H, W, torch.float16, DEV)
    x, t, y2, y1 = (pl._fat(H, W, 64, B) for _ in range(4))
    x.copy_(torch.randn(x.shape, device=DEV))
    w1 = torch.randn(64, 64, 3, 3) * (1.0 / 24.0)
    w2 = torch.randn(64, 64, 3, 3) * (1.0 / 24.0)
    b1, b2 = torch.randn(64) * 0.1, torch.randn(64) * 0.1
    seg = []
    pl.conv(seg, 'conv1', [pl.fsrc(x, 0)], [_Dst(pl.fview(t), range(64), L.ACT_RELU)], H, W, batch=B, weight=w1, bias=b1)
    pl.conv(seg, 'conv2', [pl.fsrc(t, 0)], [_Dst(pl.fview(y2), range(64), L.ACT_NONE, res=pl.fview(x))], H, W, batch=B, weight=w2, bias=b2)
    # the same block writing to y1 through the fused kernel
    pl.conv(seg, 'conv2f', [pl.fsrc(t, 0)], [_Dst(pl.fview(y1), range(64), L.ACT_NONE, res=pl.fview(x))], H, W, batch=B, weight=w2, bias=b2)
    pl._upload()
    assert pl.lib.demfi_resblock_eligible(C.byref(pl._descs[0]), C.byref(pl._descs[2])) == 1
    assert pl.lib.demfi_resblock_eligible(C.byref(pl._descs[1]), C.byref(pl._descs[2])) == 0      # not a block: no ReLU / wrong chaining
    pl.launch_resblock(0, 2, _stream())
    torch.cuda.synchronize()
    assert float(t.abs().max()) == 0.0                         # the intermediate never went to memory
    pl.launch_conv(0, _stream())
    pl.launch_conv(1, _stream())
    torch.cuda.synchronize()
    got, two = y1.float().cpu(), y2.float().cpu()
    assert torch.isfinite(got).all()
    # (ii) fp64 reference with the fp16 intermediate
    xin = x.permute(0, 3, 1, 2).double().cpu()
    m = torch.relu(torch.nn.functional.conv2d(xin, w1.half().double(), b1.double(), padding=1)).half().double()
    ref = torch.nn.functional.conv2d(m, w2.half().double(), b2.double(), padding=1) + xin
    err = (got.permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (case, err)
    # (i) the two-launch form: the same fp32 sum up to its summation order (|terms| ~ 4: a few 1e-6), i.e. the same fp16 value or its
    # neighbour where the sum sits on a rounding boundary
    d = (got - two).abs()
    ulp = torch.ldexp(torch.ones(()), torch.frexp(torch.maximum(two.abs(), torch.tensor(2.0 ** -14)))[1] - 1 - 10)   # fp16 spacing at |two|
    assert float(((d - 4e-6).clamp(min=0) / ulp).max()) <= 1.0, (case, float((d / ulp).max()))
    assert float((d > 0).float().mean()) < 0.05               # and almost everywhere identical


@pytest.mark.parametrize('H,W,q', [(16, 32, 1), (37, 75, 2), (70, 100, 3), (368, 640, 0), (368, 640, 3), (133, 530, 2)])
def test_rdb_growth_conv_streamed_weights(H, W, q):
    """Round 5: the RDB growth convolutions (DeMFInet.py:266-281: 3x3, 96 + 32 q -> 32, ReLU, written into the next 32 channels of the
    128-channel growth buffer it also reads) on the 3x3 / 32-cout instantiation of the streamed-weight kernel: 32-channel units from
    two pieces with DIFFERENT pixel strides (a 96-channel slice of the 1152-channel GFF input, the growth buffer), 32 x 32-pixel tiles,
    ragged edges, several tiles per workgroup (133 x 530), against an fp64 convolution of the same fp16 operands."""
    torch.manual_seed(H + W + q)
    pl = Plan(H, W, torch.float16, DEV)
    cat = pl._fat(H, W, 1152)
    grow = pl._fat(H, W, 128)
    cat.copy_(torch.randn(cat.shape, device=DEV) * 0.5)
    grow.copy_(torch.relu(torch.randn(grow.shape, device=DEV)))
    g0 = grow.clone()
    cin = 96 + 32 * q
    wt = torch.randn(32, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    bs = torch.randn(32) * 0.1
    srcs = [pl.fsrc(cat, 0, c0=192, nch=96)] + ([pl.fsrc(grow, 96, c0=0, nch=32 * q)] if q else [])
    pl.conv([], 'rdb', srcs, [_Dst(pl.fview(grow, 32 * q), range(32), L.ACT_RELU)], H, W, weight=wt, bias=bs)
    d = pl._descs[0]
    assert d.rec_bytes == 64 and d.cout_perm == 1 and d.n_chunks == 3 + q      # the shape the streamed-weight kernel owns
    pl._upload()
    pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    xin = torch.cat([cat[0, :, :, 192:288], g0[0, :, :, :32 * q]], 2).permute(2, 0, 1).double().cpu()[None]
    ref = torch.relu(torch.nn.functional.conv2d(xin, wt.half().double(), bs.double(), padding=1))[0]
    got = grow[0, :, :, 32 * q:32 * q + 32].permute(2, 0, 1).double().cpu()
    err = (got - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (H, W, q, err)
    keep = [c for c in range(128) if not 32 * q <= c < 32 * q + 32]
    assert torch.equal(grow[0][:, :, keep], g0[0][:, :, keep])          # nothing else of the growth buffer was touched


def test_persistent_conv_needs_its_cout_order():
    """The 64-channel 3x3 layers of the persistent kernel are packed in a permuted cout order (demfi_conv.cout_perm, set by
    demfi_conv_build): a descriptor of that shape without the flag, or a flagged one the kernel cannot take (no zero page),
    is refused instead of computing with the wrong channel order."""
    pl = Plan(16, 64, torch.float16, DEV)
    x, out = pl._fat(16, 64, 64), pl._fat(16, 64, 64)
    seg = []
    pl.conv(seg, 'p', [pl.fsrc(x, 0)], [_Dst(pl.fview(out), range(64), L.ACT_RELU)], 16, 64, weight=torch.randn(64, 64, 3, 3) * 0.05,
            bias=torch.zeros(64))
    assert pl._descs[0].cout_perm == 1
    pl._upload()
    pl.launch_conv(0, _stream())                               # fine as built
    d = pl._descs[0]
    d.cout_perm = 0
    with pytest.raises(L.DemfiError, match='cout_perm'):
        L.check(pl.lib.demfi_conv2d(C.byref(d), pl.desc_dev.data_ptr(), _stream()), 'conv')
    d.cout_perm = 1
    d.zero_page = None
    with pytest.raises(L.DemfiError, match='cout_perm'):
        L.check(pl.lib.demfi_conv2d(C.byref(d), pl.desc_dev.data_ptr(), _stream()), 'conv')
    torch.cuda.synchronize()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_conv_multi_piece_routing_upsample_shuffle(dtype):
    """Mixed fat/thin/upsampled inputs, PixelShuffle store, planar outputs with residual, GRU modes."""
    torch.manual_seed(5)
    H, W = 16, 32
    pl = Plan(H, W, dtype, DEV)
    a = pl._fat(H, W, 32)
    lo = pl._fat(H // 2, W // 2, 16)
    th = pl._thin(5)
    for t in (a, lo, th):
        t.copy_(torch.randn(t.shape, device=DEV))
    o_fat = pl._fat(H, W, 24)
    o_thin = pl._thin(3)
    r_thin = pl._thin(3)
    r_thin.copy_(torch.randn(r_thin.shape, device=DEV))
    wt = torch.randn(27, 53, 3, 3) * 0.05
    bs = torch.randn(27) * 0.1
    seg = []
    pl.conv(seg, 'mix', [pl.fsrc(a, 0), pl.tsrc(th, range(32, 37)), pl.fsrc(lo, 37, up=1)],
            [_Dst(pl.fview(o_fat), range(3, 27), L.ACT_TANH), _Dst(pl.tview(o_thin), range(0, 3), res=pl.tview(r_thin))],
            H, W, weight=wt, bias=bs)
    # PixelShuffle: 64 couts at HxW -> 16 channels at 2Hx2W
    ps_out = pl._fat(2 * H, 2 * W, 16)
    w2 = torch.randn(64, 32, 3, 3) * 0.05
    b2 = torch.randn(64) * 0.1
    pl.conv(seg, 'ps', [pl.fsrc(a, 0)],
            [_Dst(pl.fview(ps_out), [c * 4 + i * 2 + j for c in range(16)], scale=2, dy=i, dx=j) for i in range(2) for j in range(2)],
            H, W, weight=w2, bias=b2)
    # GRU pieces
    h = pl._fat(H, W, 64)
    xx = pl._fat(H, W, 64)
    h.copy_(torch.tanh(torch.randn(h.shape, device=DEV)))
    xx.copy_(torch.randn(xx.shape, device=DEV))
    zb, rh, hn = pl._fat(H, W, 64), pl._fat(H, W, 64), pl._fat(H, W, 64)
    wzr = torch.randn(128, 128, 1, 5) * 0.05
    bzr = torch.randn(128) * 0.1
    wq = torch.randn(64, 128, 1, 5) * 0.05
    bq = torch.randn(64) * 0.1
    pl.conv(seg, 'zr', [pl.fsrc(h, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(zb), range(0, 64), L.ACT_SIGMOID), _Dst(pl.fview(rh), range(64, 128), mode=L.MODE_MUL, res=pl.fview(h))],
            H, W, weight=wzr, bias=bzr)
    pl.conv(seg, 'q', [pl.fsrc(rh, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W, weight=wq, bias=bq)
    pl._upload()
    for i in range(4):
        pl.launch_conv(i, _stream())
    torch.cuda.synchronize()
    q = (lambda z: z.half().float()) if dtype == torch.float16 else (lambda z: z)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    F = torch.nn.functional
    nchw = lambda t: t[0].permute(2, 0, 1).float().cpu()[None]
    xin = torch.cat([nchw(a), q(th.cpu())[None], F.interpolate(nchw(lo), scale_factor=2, mode='nearest')], 1)
    ref = F.conv2d(xin, q(wt), bs, padding=1)[0]
    assert (o_thin.cpu() - (ref[0:3] + r_thin.cpu())).abs().max() < tol
    assert (nchw(o_fat)[0] - torch.tanh(ref[3:27])).abs().max() < tol
    ref2 = F.pixel_shuffle(F.conv2d(nchw(a), q(w2), b2, padding=1), 2)[0]
    assert (nchw(ps_out)[0] - ref2).abs().max() < tol
    hx = torch.cat([nchw(h), nchw(xx)], 1)
    zr = F.conv2d(hx, q(wzr), bzr, padding=(0, 2))
    z, r = torch.sigmoid(zr[:, :64]), torch.sigmoid(zr[:, 64:])
    assert (nchw(zb) - z).abs().max() < tol
    assert (nchw(rh) - r * nchw(h)).abs().max() < tol
    qv = torch.tanh(F.conv2d(torch.cat([nchw(rh), nchw(xx)], 1), q(wq), bq, padding=(0, 2)))
    assert (nchw(hn) - ((1 - nchw(zb)) * nchw(h) + nchw(zb) * qv)).abs().max() < tol


NARROW_CASES = [
    # pieces (channels of each NHWC input; 'p8' = 8-channel buffer of which 5 are used), cout, res, act
    ((32,), 32, False, L.ACT_RELU),            # Mixer conv_delta2
    ((32,), 64, True, L.ACT_NONE),             # conv_blend2 shape, with a residual
    ((32, 32), 32, False, L.ACT_RELU),         # conv_blend1: two pieces in one 128-byte record
    (('p8',), 32, False, L.ACT_RELU),          # conv_delta1: 5 of 8 channels, record padded to one k-step
    ((16,), 64, False, L.ACT_NONE),
    ((48, 16), 64, True, L.ACT_RELU),
    ((16, 8), 32, False, L.ACT_NONE),          # 48-byte chunk padded to 64
]


@pytest.mark.parametrize('case', NARROW_CASES)
@pytest.mark.parametrize('H,W,batch', [(8, 32, 1), (37, 75, 2), (64, 96, 1), (19, 130, 1)])
def test_narrow_persistent_conv(case, H, W, batch):
    """3x3 layers with <= 64 input channels (Mixer, DeMFInet.py:800-836) through the narrow persistent kernel:
    record sizes 32 / 64 / 128 B, two-piece records, zero-padded records, ragged tile edges, batch > 1."""
    pieces, cout, res, act = case
    torch.manual_seed(3)
    pl = Plan(H, W, torch.float16, DEV)
    srcs, xs, cin = [], [], 0
    for pc in pieces:
        if pc == 'p8':
            b = pl._fat(H, W, 8, batch)
            b.copy_(torch.randn(b.shape, device=DEV))
            srcs.append(pl.fsrc_map(b, [cin + i for i in range(5)] + [-1] * 3, b=None))     # batch stride kept: the batched plan runs this shape at batch 7
            xs.append(b[..., :5])
            cin += 5
        else:
            b = pl._fat(H, W, pc, batch)
            b.copy_(torch.randn(b.shape, device=DEV))
            srcs.append(pl.fsrc(b, cin))
            xs.append(b)
            cin += pc
    out = pl._fat(H, W, cout, batch)
    r = pl._fat(H, W, cout, batch) if res else None
    if res:
        r.copy_(torch.randn(r.shape, device=DEV))
    wt = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    bs = torch.randn(cout) * 0.1
    seg = []
    pl.conv(seg, 'narrow', srcs, [_Dst(pl.fview(out), range(cout), act, res=pl.fview(r) if res else None)], H, W, batch=batch,
            weight=wt, bias=bs)
    pl._upload()
    for rep in range(2):
        out.zero_()
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    F = torch.nn.functional
    nchw = lambda t: t.permute(0, 3, 1, 2).double().cpu()
    ref = F.conv2d(torch.cat([nchw(x) for x in xs], 1), wt.half().double(), bs.double(), padding=1)
    if res:
        ref = ref + nchw(r)
    if act == L.ACT_RELU:
        ref = torch.relu(ref)
    err = (nchw(out) - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (case, err)


@pytest.mark.parametrize('H,W', [(8, 32), (37, 75), (64, 96), (19, 130)])
def test_narrow_persistent_conv_7x7(H, W):
    """Mixer.conv_delta1 (7x7, 5 -> 32, DeMFInet.py:800-812): the 7x7 instantiation of the narrow persistent kernel (49 taps of
    resident weights, one 16-channel k-step per tap, 14x38-pixel tiles) instead of the general kernel's 49 per-tap barriers.  Round 6: the chunk
    [8-channel piece | 8 zero channels] of this test (and of the plan) takes the PAIRED-TAP mode -- the upper-half lanes of the B operand read the next
    column, one MFMA covers the taps (ky, 2j) and (ky, 2j + 1): 28 k-steps instead of 49, four DMA waves (DEMFI_N7_PAIR=0: one tap per k-step)."""
    torch.manual_seed(5)
    pl = Plan(H, W, torch.float16, DEV)
    b = pl._fat(H, W, 8)
    b.copy_(torch.randn(b.shape, device=DEV))
    out = pl._fat(H, W, 32)
    wt = torch.randn(32, 5, 7, 7) * (1.0 / (5 * 49) ** 0.5)
    bs = torch.randn(32) * 0.1
    pl.conv([], 'delta1', [pl.fsrc_map(b, [0, 1, 2, 3, 4, -1, -1, -1])], [_Dst(pl.fview(out), range(32), L.ACT_RELU)], H, W, weight=wt, bias=bs)
    pl._upload()
    for rep in range(2):
        out.zero_()
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    nchw = lambda t: t.permute(0, 3, 1, 2).double().cpu()
    ref = torch.relu(torch.nn.functional.conv2d(nchw(b[..., :5]), wt.half().double(), bs.double(), padding=3))
    err = (nchw(out) - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('H,W,batch,act', [(16, 32, 1, L.ACT_TANH), (37, 75, 2, L.ACT_NONE), (64, 96, 1, L.ACT_RELU), (19, 130, 3, L.ACT_TANH),
                                           (48, 40, 1, L.ACT_NONE),
                                           (152, 420, 2, L.ACT_TANH)])    # 10 x 14 x 2 = 280 tiles > 256 workgroups: XCD bands, several tiles per workgroup
def test_streamed_weight_conv_7x7_192_to_64(H, W, batch, act):
    """Ch_Reducer (7x7, 3 x 64 -> 64, DeMFInet.py:37, 114): the streamed-weight kernel (16 x 32-pixel tiles, 8 accumulators per wave,
    A fragments straight from the packed weights in L2, 32-channel activation units through a double buffer) against torch on
    tiles that are interior, ragged at both edges, several per workgroup sequence, and with a batch stride.  The three pieces
    are channel slices of ONE 192-channel NHWC buffer, as in the plan."""
    torch.manual_seed(11)
    pl = Plan(H, W, torch.float16, DEV)
    b = pl._fat(H, W, 192, batch)
    b.copy_(torch.randn(b.shape, device=DEV))
    out = pl._fat(H, W, 64, batch)
    wt = torch.randn(64, 192, 7, 7) * (1.0 / (192 * 49) ** 0.5)
    bs = torch.randn(64) * 0.1
    pl.conv([], 'Ch_Reducer', [pl.fsrc(b, 0, 0, 64), pl.fsrc(b, 64, 64, 64), pl.fsrc(b, 128, 128, 64)],
            [_Dst(pl.fview(out), range(64), act)], H, W, batch=batch, weight=wt, bias=bs)
    assert pl._descs[0].cout_perm == 1, 'the layer must be packed for the streamed-weight kernel'
    pl._upload()
    for rep in range(2):
        out.fill_(7.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    nchw = lambda t: t.permute(0, 3, 1, 2).double().cpu()
    big = H * W * batch > 50000                                  # the multi-tile case: fp32 reference on the GPU (a double conv on the CPU takes a minute)
    if big:
        ref = torch.nn.functional.conv2d(b.permute(0, 3, 1, 2).float(), wt.half().float().to(DEV), bs.to(DEV), padding=3).double().cpu()
    else:
        ref = torch.nn.functional.conv2d(nchw(b), wt.half().double(), bs.double(), padding=3)
    ref = {L.ACT_TANH: torch.tanh, L.ACT_RELU: torch.relu, L.ACT_NONE: (lambda x: x)}[act](ref)
    err = (nchw(out) - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


WS2_S2_CASES = [
    # (input pieces as (channels of the buffer, channels taken), cout, residual?, act)
    ([(64, 64), (16, 16)], 64, True, L.ACT_RELU),                # Refine_Module.enc1#t: Ft | the 16-channel record of the thin planes, + the hoisted aF part
    ([(64, 64), (64, 64)], 64, False, L.ACT_NONE),               # enc1#aF
    ([(64, 64)], 128, False, L.ACT_RELU),                        # enc2: two 64-cout blocks
    ([(128, 128)], 256, False, L.ACT_RELU),                      # enc3: four
]


@pytest.mark.parametrize('case', WS2_S2_CASES)
@pytest.mark.parametrize('H,W,batch', [(16, 32, 1), (37, 75, 2), (23, 40, 1), (184, 320, 2)])
def test_stride2_4x4_conv_by_phases(case, H, W, batch):
    """Round 6 (wsconv.hip): the 4x4 stride-2 layers of the UNet encoder (DeMFInet.py:575-577, 588-590) as four phases of 2x2 taps over
    32-channel units -- every tap of every phase against torch's strided convolution (fp64 on the same fp16 operands): interior and
    ragged tiles, odd output sizes (the input is 2H x 2W), a 16-channel tail piece padded to a unit, several 64-cout blocks, a batch
    stride, the NHWC residual, more items than workgroups (184 x 320 x 2)."""
    pieces, cout, with_res, act = case
    if H * W * batch > 50000 and cout > 64:
        pytest.skip('the large grid is covered by the 64-cout cases')
    torch.manual_seed(H * 7 + W + cout)
    pl = Plan(H, W, torch.float16, DEV)
    bufs, srcs, cin = [], [], 0
    for ct, take in pieces:
        b = pl._fat(2 * H, 2 * W, ct, batch)
        b.copy_(torch.randn(b.shape, device=DEV))
        bufs.append((b, take))
        if take == 16:
            m = list(range(cin, cin + 9)) + [-1] * 7             # 9 real channels of a 16-channel record (misc16)
            srcs.append(pl.fsrc_map(b, m, b=None))
            cin += 9
        else:
            srcs.append(pl.fsrc(b, cin, 0, take))
            cin += take
    out = pl._fat(H, W, cout, batch)
    res = pl._fat(H, W, cout, batch)
    res.copy_(torch.randn(res.shape, device=DEV))
    wt = torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5)
    bs = torch.randn(cout) * 0.1
    pl.conv([], 'enc', srcs, [_Dst(pl.fview(out), range(cout), act, res=pl.fview(res) if with_res else None)], H, W, stride=2, batch=batch,
            weight=wt, bias=bs)
    d = pl._descs[0]
    assert d.rec_bytes == 64 and d.nco == 2 and d.cout_perm == 1 and d.n_chunks == sum((t + 31) // 32 for _, t in pieces)      # the shape wsconv.hip owns
    pl._upload()
    for rep in range(2):
        out.fill_(7.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    xs = []
    for b, take in bufs:
        xs.append(b[..., :9] if take == 16 else b[..., :take])
    x = torch.cat(xs, 3).permute(0, 3, 1, 2)
    big = H * W * batch > 50000
    if big:
        ref = torch.nn.functional.conv2d(x.float(), wt.half().float().to(DEV), bs.to(DEV), stride=2, padding=1).double().cpu()
    else:
        ref = torch.nn.functional.conv2d(x.double().cpu(), wt.half().double(), bs.double(), stride=2, padding=1)
    if with_res:
        ref = ref + res.permute(0, 3, 1, 2).double().cpu()
    if act == L.ACT_RELU:
        ref = torch.relu(ref)
    got = out.permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (case, H, W, batch, err)


WS2_S1_CASES = [
    # (pieces as (channels, upsampled?), cout, residual?, act)
    ([(128, True), (64, False)], 64, False, L.ACT_RELU),         # Refine_Module.dec2: cat[up(d1), u1]
    ([(256, True), (128, False)], 128, False, L.ACT_RELU),       # dec1: two 64-cout blocks
    ([(256, False)], 256, False, L.ACT_RELU),                    # dec0
    ([(64, False), (64, False)], 64, False, L.ACT_RELU),         # FGAC w_gen
    ([(64, False), (32, False)], 64, True, L.ACT_RELU),          # 96 channels + NHWC residual (the shape of a fused Dec_first_2)
]


@pytest.mark.parametrize('case', WS2_S1_CASES)
@pytest.mark.parametrize('H,W,batch', [(16, 32, 1), (38, 76, 2), (22, 40, 1), (184, 320, 2)])
def test_conv3x3_over_32_channel_units(case, H, W, batch):
    """Round 6 (wsconv.hip), the 3x3 form: K walked in 32-channel units from several NHWC pieces, some of them read through the x2
    nearest-neighbour upsample of the UNet decoder (DeMFInet.py:592-601: cat[up(d), skip]), 64-cout blocks, optional residual --
    against torch (upsample + conv2d, fp64 on the same fp16 operands) on interior / ragged tiles and more items than workgroups."""
    pieces, cout, with_res, act = case
    if H * W * batch > 50000 and cout > 64:
        pytest.skip('the large grid is covered by the 64-cout cases')
    torch.manual_seed(H * 5 + W + cout)
    pl = Plan(H, W, torch.float16, DEV)
    srcs, xs, cin = [], [], 0
    for ch, up in pieces:
        b = pl._fat(H // 2, W // 2, ch, batch) if up else pl._fat(H, W, ch, batch)
        b.copy_(torch.randn(b.shape, device=DEV))
        srcs.append(pl.fsrc(b, cin, 0, ch, up=1 if up else 0))
        x = b.permute(0, 3, 1, 2)
        xs.append(torch.nn.functional.interpolate(x.float(), scale_factor=2, mode='nearest').half() if up else x)
        cin += ch
    out = pl._fat(H, W, cout, batch)
    res = pl._fat(H, W, cout, batch)
    res.copy_(torch.randn(res.shape, device=DEV))
    wt = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    bs = torch.randn(cout) * 0.1
    pl.conv([], 'dec', srcs, [_Dst(pl.fview(out), range(cout), act, res=pl.fview(res) if with_res else None)], H, W, batch=batch, weight=wt, bias=bs)
    d = pl._descs[0]
    assert d.rec_bytes == 64 and d.nco == 2 and d.cout_perm == 1 and d.n_chunks == cin // 32
    pl._upload()
    for rep in range(2):
        out.fill_(7.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    x = torch.cat(xs, 1)
    if H * W * batch > 50000:
        ref = torch.nn.functional.conv2d(x.float(), wt.half().float().to(DEV), bs.to(DEV), padding=1).double().cpu()
    else:
        ref = torch.nn.functional.conv2d(x.double().cpu(), wt.half().double(), bs.double(), padding=1)
    if with_res:
        ref = ref + res.permute(0, 3, 1, 2).double().cpu()
    ref = torch.relu(ref)
    err = (out.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (case, H, W, batch, err)


@pytest.mark.parametrize('H,W,batch', [(16, 32, 1), (37, 75, 2), (184, 320, 2)])
def test_conv3x3_units_with_a_two_piece_tail(H, W, batch):
    """The fused Dec_first_2 of the fp16 plan (DeMFInet.py:151-158, round 6): units [F_rec lo | F_rec hi | 16 channels of ref16 through a
    channel map + the 8-channel record of the recursion + zero padding], the window-constant partial sum as a residual WITHOUT batch
    stride (one image shared by all time instants), ReLU -- against torch on the same fp16 operands."""
    torch.manual_seed(H + W)
    pl = Plan(H, W, torch.float16, DEV)
    h = pl._fat(H, W, 64, batch)
    r16 = pl._fat(H, W, 16, batch)
    a8 = pl._fat(H, W, 8, batch)
    for b in (h, r16, a8):
        b.copy_(torch.randn(b.shape, device=DEV))
    gw = pl._fat(H, W, 64, 1)
    gw.copy_(torch.randn(gw.shape, device=DEV))
    out = pl._fat(H, W, 64, batch)
    m16 = [64, 65, 66, 67, 68, 69, -1, -1, -1, 71, 72, 73, 74, -1, 70, -1]
    cin = 83
    wt = torch.randn(64, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    bs = torch.randn(64) * 0.1
    res = pl.fview(gw)
    res.sb = 0
    pl.conv([], 'Dec_first_2#t', [pl.fsrc(h, 0), pl.fsrc_map(r16, m16, b=None), pl.fsrc_map(a8, list(range(75, 83)), b=None)],
            [_Dst(pl.fview(out), range(64), L.ACT_RELU, res=res)], H, W, batch=batch, weight=wt, bias=bs)
    d = pl._descs[0]
    assert d.rec_bytes == 64 and d.nco == 2 and d.cout_perm == 1 and d.n_chunks == 3 and d.chunks[2].n_pieces == 3
    pl._upload()
    for rep in range(2):
        out.fill_(7.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    x = torch.zeros(batch, cin, H, W, dtype=torch.float16, device=DEV)
    x[:, :64] = h.permute(0, 3, 1, 2)
    for ch, ci in enumerate(m16):
        if ci >= 0:
            x[:, ci] = r16[..., ch]
    x[:, 75:83] = a8.permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x.float(), wt.half().float().to(DEV), bs.to(DEV), padding=1) + gw.permute(0, 3, 1, 2).float()
    ref = torch.relu(ref).double().cpu()
    err = (out.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), (H, W, batch, err)


THIN_CASES = [
    # cin, list of (n couts, residual?) per destination tensor, act, batch
    (64, [(3, True), (3, True), (3, True)], L.ACT_NONE, 1),      # Dec_last2_2: three frames, each + its own residual
    (64, [(3, False)], L.ACT_NONE, 3),                           # Dec_last2: batch 3 with a batch stride on the planes
    (32, [(5, True)], L.ACT_NONE, 1),                            # flow_occ.conv2: 5 channels straddle the two lane halves
    (64, [(1, False)], L.ACT_SIGMOID, 1),                        # w_gen_2
    (16, [(8, True), (2, False)], L.ACT_TANH, 1),
]


@pytest.mark.parametrize('case', THIN_CASES)
@pytest.mark.parametrize('H,W', [(8, 32), (37, 75), (64, 96)])
def test_narrow_persistent_conv_thin_outputs(case, H, W):
    """3x3 layers writing planar fp32 outputs (frames, flow/occlusion deltas, gates) through the THIN epilogue of the narrow
    persistent kernel: per-octet routing to several tensors, planar residuals, batch stride, ragged edges."""
    cin, dsts, act, batch = case
    torch.manual_seed(17)
    pl = Plan(H, W, torch.float16, DEV)
    x = pl._fat(H, W, cin, batch)
    x.copy_(torch.randn(x.shape, device=DEV))
    outs, ress, D, c0 = [], [], [], 0
    for n, has_res in dsts:
        o = torch.zeros((batch * n, H, W), dtype=torch.float32, device=DEV)
        r = torch.randn((batch * n, H, W), dtype=torch.float32, device=DEV) if has_res else None
        outs.append(o)
        ress.append(r)
        sb = n * H * W if batch > 1 else 0
        D.append(_Dst(pl.tview(o, 0, sb=sb), range(c0, c0 + n), act, res=pl.tview(r, 0, sb=sb) if has_res else None))
        c0 += n
    cout = c0
    wt = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    bs = torch.randn(cout) * 0.1
    pl.conv([], 'thin', [pl.fsrc(x, 0)], D, H, W, batch=batch, weight=wt, bias=bs)
    pl._upload()
    for rep in range(2):
        for o in outs:
            o.fill_(-77.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    F = torch.nn.functional
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.half().double(), bs.double(), padding=1)
    fn = {L.ACT_NONE: lambda z: z, L.ACT_TANH: torch.tanh, L.ACT_SIGMOID: torch.sigmoid}[act]
    c0 = 0
    for (n, has_res), o, r in zip(dsts, outs, ress):
        e = ref[:, c0:c0 + n]
        if has_res:
            e = e + r.view(batch, n, H, W).double().cpu()
        e = fn(e)
        got = o.view(batch, n, H, W).double().cpu()
        err = (got - e).abs().max().item()
        assert err < 4e-3 * max(1.0, e.abs().max().item()), (case, err)
        c0 += n


@pytest.mark.parametrize('dsts,pack_ch', [([(5, True)], [0]), ([(4, True), (1, True)], [0, 4]), ([(3, False), (5, True)], [-1, 8])])
@pytest.mark.parametrize('H,W,batch', [(8, 32, 1), (37, 75, 2), (64, 96, 1)])
def test_thin_outputs_with_packed_copy(dsts, pack_ch, H, W, batch):
    """demfi_conv.pack (ABI v6): the thin epilogue also writes its planes as fp16 channels of an NHWC record -- exactly what a
    demfi_pack_planes launch over the fp32 planes produces (the per-recursion pack of the flow / occlusion deltas,
    DeMFInet.py:130-137 -> Mixer.conv_delta1, is gone from the plan).  Shapes: flow_occ.conv2 (5 channels straddling the lane
    halves), dec3's plane launches (4 + 1 in two octets), an octet that is not packed; channels nobody writes keep their value."""
    torch.manual_seed(23)
    pl = Plan(H, W, torch.float16, DEV)
    x = pl._fat(H, W, 32, batch)
    x.copy_(torch.randn(x.shape, device=DEV))
    rec = pl._fat(H, W, 16, batch)
    outs, ress, D, c0 = [], [], [], 0
    for n, has_res in dsts:
        o = torch.zeros((batch * n, H, W), dtype=torch.float32, device=DEV)
        r = torch.randn((batch * n, H, W), dtype=torch.float32, device=DEV) if has_res else None
        outs.append(o)
        ress.append(r)                                   # the descriptor holds a raw pointer: the residual must stay alive
        sb = n * H * W if batch > 1 else 0
        D.append(_Dst(pl.tview(o, 0, sb=sb), range(c0, c0 + n), L.ACT_NONE, res=pl.tview(r, 0, sb=sb) if has_res else None))
        c0 += n
    wt = torch.randn(c0, 32, 3, 3) * (1.0 / (32 * 9) ** 0.5)
    pl.conv([], 'thinpack', [pl.fsrc(x, 0)], D, H, W, batch=batch, weight=wt, bias=torch.randn(c0) * 0.1, pack=(pl.fview(rec), pack_ch))
    pl._upload()
    for rep in range(2):
        rec.fill_(-3.0)
        pl.launch_conv(0, _stream())
    torch.cuda.synchronize()
    exp = torch.full((batch, H, W, 16), -3.0, dtype=torch.float16)
    for (n, _), o, chn in zip(dsts, outs, pack_ch):
        if chn < 0:
            continue
        n4 = (n + 3) // 4 * 4
        exp[..., chn:chn + n4] = 0.0
        exp[..., chn:chn + n] = o.view(batch, n, H, W).permute(0, 2, 3, 1).half().cpu()    # the pack kernel's conversion of the stored planes
        assert torch.isfinite(o).all()
    assert torch.equal(rec.cpu(), exp)


@pytest.mark.parametrize('kh,kw', [(1, 5), (5, 1)])
@pytest.mark.parametrize('H,W,batch', [(8, 32, 1), (37, 75, 2), (64, 96, 1), (100, 45, 1), (33, 8, 3)])
def test_sep_gru_persistent_kernel(kh, kw, H, W, batch):
    """SepConvGRU half-step (DeMFInet.py:844-849 / 851-856) through the persistent 1x5 / 5x1 kernel: fused z|r launch
    (sigmoid, sigmoid*h) and the q launch ((1-z)h + z tanh), ragged tile edges, batch > 1, both orientations."""
    torch.manual_seed(11 + kh)
    pl = Plan(H, W, torch.float16, DEV)
    h, xx = pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch)
    h.copy_(torch.tanh(torch.randn(h.shape, device=DEV)))
    xx.copy_(torch.randn(xx.shape, device=DEV))
    zb, rh, hn = pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch)
    wzr = torch.randn(128, 128, kh, kw) * 0.05
    bzr = torch.randn(128) * 0.1
    wq = torch.randn(64, 128, kh, kw) * 0.05
    bq = torch.randn(64) * 0.1
    seg = []
    pl.conv(seg, 'zr', [pl.fsrc(h, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(zb), range(0, 64), L.ACT_SIGMOID), _Dst(pl.fview(rh), range(64, 128), mode=L.MODE_MUL, res=pl.fview(h))],
            H, W, batch=batch, weight=wzr, bias=bzr)
    pl.conv(seg, 'q', [pl.fsrc(rh, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W, batch=batch,
            weight=wq, bias=bq)
    pl._upload()
    for rep in range(2):                                   # twice: the launch leaves no state behind
        zb.zero_(); rh.zero_(); hn.zero_()
        pl.launch_conv(0, _stream())
        pl.launch_conv(1, _stream())
    torch.cuda.synchronize()
    F = torch.nn.functional
    nchw = lambda t: t.permute(0, 3, 1, 2).float().cpu()
    q16 = lambda z: z.half().float()
    pad = (kh // 2, kw // 2)
    zr = F.conv2d(torch.cat([nchw(h), nchw(xx)], 1), q16(wzr), bzr, padding=pad)
    z, r = torch.sigmoid(zr[:, :64]), torch.sigmoid(zr[:, 64:])
    assert (nchw(zb) - z).abs().max() < 1e-2
    assert (nchw(rh) - r * nchw(h)).abs().max() < 1e-2
    qv = torch.tanh(F.conv2d(torch.cat([nchw(rh), nchw(xx)], 1), q16(wq), bq, padding=pad))
    assert (nchw(hn) - ((1 - nchw(zb)) * nchw(h) + nchw(zb) * qv)).abs().max() < 1e-2


@pytest.mark.parametrize('kh,kw', [(1, 5), (5, 1)])
@pytest.mark.parametrize('H,W,batch', [(8, 32, 1), (37, 75, 2), (64, 96, 1), (100, 45, 1), (33, 8, 3), (16, 160, 1), (96, 64, 2)])
def test_gru_half_step_r_then_zq(kh, kw, H, W, batch):
    """Round 6 (gru.hip): a SepConvGRU half-step (DeMFInet.py:844-849 / 851-856) as  r*h  (demfi_gru_r) and  z + q + blend  in one launch
    (demfi_gru_zq; z stays on chip).  Checked against fp32 torch on the same fp16 operands and against the round-5 path (the same three
    layers through demfi_conv2d: equal up to the summation order of the fp32 accumulators).  Ragged tiles in both directions, batch > 1,
    tiles that straddle the image along the filter axis (lines) and across it (pixels)."""
    torch.manual_seed(23 + kh)
    pl = Plan(H, W, torch.float16, DEV)
    h, xx = pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch)
    h.copy_(torch.tanh(torch.randn(h.shape, device=DEV)))
    xx.copy_(torch.randn(xx.shape, device=DEV))
    zb, rh, hn = pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch)
    rh2, hn2 = pl._fat(H, W, 64, batch), pl._fat(H, W, 64, batch)
    wz, wr, wq = (torch.randn(64, 128, kh, kw) * 0.05 for _ in range(3))
    bz, br, bq = (torch.randn(64) * 0.1 for _ in range(3))
    seg = []
    pl.conv(seg, 'r', [pl.fsrc(h, 0), pl.fsrc(xx, 64)], [_Dst(pl.fview(rh), range(64), mode=L.MODE_MUL, res=pl.fview(h))],
            H, W, batch=batch, weight=wr, bias=br)
    pl.conv(seg, 'z', [pl.fsrc(h, 0), pl.fsrc(xx, 64)], [_Dst(pl.fview(zb), range(64), L.ACT_SIGMOID)], H, W, batch=batch, weight=wz, bias=bz)
    pl.conv(seg, 'q', [pl.fsrc(rh, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W, batch=batch, weight=wq, bias=bq)
    # the same layers writing other buffers: the round-5 path (demfi_conv2d)
    pl.conv(seg, 'r', [pl.fsrc(h, 0), pl.fsrc(xx, 64)], [_Dst(pl.fview(rh2), range(64), mode=L.MODE_MUL, res=pl.fview(h))],
            H, W, batch=batch, weight=wr, bias=br)
    pl.conv(seg, 'q', [pl.fsrc(rh2, 0), pl.fsrc(xx, 64)],
            [_Dst(pl.fview(hn2), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W, batch=batch, weight=wq, bias=bq)
    pl._upload()
    assert pl.lib.demfi_gru_r_eligible(C.byref(pl._descs[0])) == 1
    assert pl.lib.demfi_gru_zq_eligible(C.byref(pl._descs[1]), C.byref(pl._descs[2])) == 1
    assert pl.lib.demfi_gru_zq_eligible(C.byref(pl._descs[0]), C.byref(pl._descs[2])) == 0          # r is not an update gate
    assert pl.lib.demfi_gru_r_eligible(C.byref(pl._descs[1])) == 0
    for rep in range(2):                                   # twice: the launches leave no state behind
        zb.fill_(7.0); rh.zero_(); hn.zero_()
        pl.launch_gru_r(0, _stream())
        pl.launch_gru_zq(1, 2, _stream())
    torch.cuda.synchronize()
    assert float(zb.min()) == 7.0                           # the fused launch never touches the z buffer
    F = torch.nn.functional
    nchw = lambda t: t.permute(0, 3, 1, 2).float().cpu()
    q16 = lambda z: z.half().float()
    pad = (kh // 2, kw // 2)
    hx = torch.cat([nchw(h), nchw(xx)], 1)
    r = torch.sigmoid(F.conv2d(hx, q16(wr), br, padding=pad))
    assert (nchw(rh) - r * nchw(h)).abs().max() < 4e-3
    z = q16(torch.sigmoid(F.conv2d(hx, q16(wz), bz, padding=pad)))                       # z is handed over as fp16
    qv = torch.tanh(F.conv2d(torch.cat([nchw(rh), nchw(xx)], 1), q16(wq), bq, padding=pad))
    assert (nchw(hn) - ((1 - z) * nchw(h) + z * qv)).abs().max() < 4e-3
    # against the round-5 launches on the same operands
    pl.launch_conv(1, _stream())                            # z -> zb
    pl.launch_conv(3, _stream())                            # r*h -> rh2
    pl.launch_conv(4, _stream())                            # -> hn2
    torch.cuda.synchronize()
    assert (rh.float() - rh2.float()).abs().max() < 2e-3
    assert (hn.float() - hn2.float()).abs().max() < 4e-3


# ------------------------------------------------------------------------------------------------------
# warps: fixtures from the reference + bit-identical integer maps
# ------------------------------------------------------------------------------------------------------
FAMS = ['zeros', 'ints', 'halves', 'smooth', 'large', 'edges', 'collide']


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _warp_maps_expected(flo, H, W):
    m = O.backward_warp_maps(flo)
    inb = sum((m['inb'][k].astype(np.int32) << k) for k in range(4)) | (m['valid'].astype(np.int32) << 4)
    x0 = np.clip(m['ix0'], -4, W + 4).astype(np.int32)
    y0 = np.clip(m['iy0'], -4, H + 4).astype(np.int32)
    return x0, y0, inb


def test_warp_blend_thin_vs_reference_goldens(golden_dir):
    g = _gold(golden_dir, 'warps_24x40')
    H, W = 24, 40
    lib = L.load()
    img = torch.from_numpy(g['img3']).to(DEV)
    t = torch.tensor([0.375], device=DEV)
    for fa_name, fb_name in zip(FAMS, FAMS[1:] + FAMS[:1]):
        fa = torch.from_numpy(g['flo_' + fa_name]).to(DEV)
        fb = torch.from_numpy(g['flo_' + fb_name]).to(DEV)
        logit = (torch.arange(H * W, device=DEV).float().view(H, W) % 7 - 3.0).contiguous()
        out = torch.zeros(3, H, W, device=DEV)
        occ = torch.zeros(H, W, device=DEV)
        dbg = torch.zeros(2, 3, H * W, dtype=torch.int32, device=DEV)
        A, B, Ov = _view_planar(img), _view_planar(img), _view_planar(out)
        L.check(lib.demfi_warp_blend(C.byref(A), fa.data_ptr(), C.byref(B), fb.data_ptr(), logit.data_ptr(), t.data_ptr(),
                                     C.byref(Ov), 3, H, W, occ.data_ptr(), dbg.data_ptr(), _stream()))
        torch.cuda.synchronize()
        # values: Eq.(2) of the reference bwarp outputs
        wa = torch.from_numpy(g['bwarp3_' + fa_name])
        wb = torch.from_numpy(g['bwarp3_' + fb_name])
        o0 = torch.sigmoid(logit.cpu())
        ref = (0.625 * o0 * wa + 0.375 * (1 - o0) * wb) / (0.625 * o0 + 0.375 * (1 - o0))
        assert (out.cpu() - ref).abs().max() < 2e-6
        assert (occ.cpu() - o0).abs().max() < 1e-6
        # integer maps: bit-identical to the step-by-step fp32 emulation that equals torch
        for which, nm in ((0, fa_name), (1, fb_name)):
            x0, y0, inb = _warp_maps_expected(g['flo_' + nm], H, W)
            got = dbg[which].cpu().numpy().reshape(3, H, W)
            assert np.array_equal(got[2], inb), nm
            anyin = (inb & 15) != 0
            assert np.array_equal(got[0][anyin], x0[anyin]) and np.array_equal(got[1][anyin], y0[anyin]), nm


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_warp_blend_fat_vs_oracle(dtype):
    torch.manual_seed(3)
    H, W, Cc = 40, 72, 64
    lib = L.load()
    A = torch.tanh(torch.randn(H, W, Cc, device=DEV)).to(dtype)
    B = torch.tanh(torch.randn(H, W, Cc, device=DEV)).to(dtype)
    fl = (torch.randn(4, H, W, device=DEV) * 6).contiguous()
    logit = torch.randn(H, W, device=DEV) * 3
    t = torch.tensor([0.25], device=DEV)
    out = torch.zeros(H, W, Cc, device=DEV, dtype=dtype)
    dbg = torch.zeros(2, 3, H * W, dtype=torch.int32, device=DEV)
    va, vb, vo = _view_nhwc(A), _view_nhwc(B), _view_nhwc(out)
    L.check(lib.demfi_warp_blend(C.byref(va), fl.data_ptr(), C.byref(vb), fl[2:].data_ptr(), logit.data_ptr(), t.data_ptr(),
                                 C.byref(vo), Cc, H, W, None, dbg.data_ptr(), _stream()))
    torch.cuda.synchronize()
    nchw = lambda z: z.permute(2, 0, 1).float().cpu()[None]
    ref = O.warp_blend(nchw(A), fl[None, 0:2].cpu(), nchw(B), fl[None, 2:4].cpu(), logit.cpu()[None, None], t.cpu().view(1, 1, 1, 1))
    tol = 3e-6 if dtype == torch.float32 else 2e-3
    assert (nchw(out) - ref).abs().max() < tol
    for which in range(2):
        x0, y0, inb = _warp_maps_expected(fl[2 * which:2 * which + 2].cpu().numpy(), H, W)
        got = dbg[which].cpu().numpy().reshape(3, H, W)
        assert np.array_equal(got[2], inb)


def test_cfr_vs_reference_goldens(golden_dir):
    g = _gold(golden_dir, 'warps_24x40')
    H, W = 24, 40
    lib = L.load()
    for i in range(4):
        f01 = torch.from_numpy(g['cfr%d_f01' % i]).to(DEV)
        f10 = torch.from_numpy(g['cfr%d_f10' % i]).to(DEV)
        tv = float(g['cfr%d_t' % i])
        t = torch.tensor([tv], device=DEV)
        acc = torch.zeros(lib.demfi_cfr_workspace_bytes(H, W) // 8, dtype=torch.int64, device=DEV)
        out = torch.zeros(4, H, W, device=DEV)
        dbg = torch.zeros(2, 4, H * W, dtype=torch.int32, device=DEV)
        L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), t.data_ptr(), H, W, acc.data_ptr(), out.data_ptr(),
                                         dbg.data_ptr(), _stream()))
        torch.cuda.synchronize()
        ref = np.concatenate([g['cfr%d_ft0' % i], g['cfr%d_ft1' % i]], 0)
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(out.cpu().numpy() - ref).max() < 2e-5 * scale, i
        # splat target indices: bit-identical to sample_one's idxx/idxy/mask
        t32 = np.float32(tv)
        for k, (fl, s) in enumerate(((g['cfr%d_f01' % i], t32), (g['cfr%d_f10' % i], np.float32(1) - t32))):
            maps = O.splat_maps((fl * s).astype(np.float32), H, W)
            for c, m in enumerate(maps):
                exp = np.where(m['mask'], m['row'] * W + m['col'], -1).astype(np.int32).reshape(-1)
                assert np.array_equal(dbg[k, c].cpu().numpy(), exp), (i, k, c)


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_cfr_writes_the_packed_record_of_the_next_layer(dtype):
    """Round 6: demfi_cfr_flow_align_pack -- the finish also writes the NHWC record [flow_t0, flow_t1 | flow_01, flow_10, logit | 7 zeros] that
    Refine_Module.enc1 stages (the thin members of Agg1, DeMFInet.py:77): bit-identical to demfi_pack_planes over the same nine planes, the
    planar outputs identical to the plain entry point; single launch and a batched one (two contexts with their own t and strides)."""
    torch.manual_seed(5)
    H, W = 72, 136
    lib = L.load()
    dt = L.F32 if dtype == torch.float32 else L.F16
    f01 = (torch.randn(2, H, W, device=DEV) * 7).contiguous()
    f10 = (torch.randn(2, H, W, device=DEV) * 7).contiguous()
    logit = torch.randn(H, W, device=DEV)
    nacc = lib.demfi_cfr_workspace_bytes(H, W) // 8
    for nb in (1, 2):
        t = torch.tensor([0.375, 0.75], device=DEV)[:nb].contiguous()
        acc = torch.zeros(nb, nacc, dtype=torch.int64, device=DEV)
        out = torch.zeros(nb, 4, H, W, device=DEV)
        rec = torch.full((nb, H, W, 16), -3.0, dtype=dtype, device=DEV)
        bt = L.Batch()
        bt.nb = nb
        bt.t = 4
        bt.p[2], bt.p[3], bt.p[5] = nacc * 8, 4 * H * W * 4, H * W * 16 * rec.element_size()      # flows and logit: window-level (stride 0)
        L.check(lib.demfi_cfr_flow_align_pack(f01.data_ptr(), f10.data_ptr(), logit.data_ptr(), t.data_ptr(), H, W, acc.data_ptr(), out.data_ptr(),
                                              rec.data_ptr(), dt, C.byref(bt) if nb > 1 else None, _stream()))
        torch.cuda.synchronize()
        assert int(acc.abs().max()) == 0                           # the workspace is left all-zero
        for q in range(nb):
            acc1 = torch.zeros(nacc, dtype=torch.int64, device=DEV)
            out1 = torch.zeros(4, H, W, device=DEV)
            L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), t[q:q + 1].data_ptr(), H, W, acc1.data_ptr(), out1.data_ptr(), None, _stream()))
            assert torch.equal(out[q], out1)
            exp = torch.full((H, W, 16), -3.0, dtype=dtype, device=DEV)
            planes = [out1[i] for i in range(4)] + [f01[0], f01[1], f10[0], f10[1], logit]
            ptrs = (C.c_void_p * 16)(*([p_.data_ptr() for p_ in planes] + [None] * 7))
            L.check(lib.demfi_pack_planes(ptrs, 16, exp.data_ptr(), dt, 16, H, W, _stream()))
            torch.cuda.synchronize()
            assert torch.equal(rec[q], exp)


def test_cfr_is_deterministic():
    torch.manual_seed(1)
    H, W = 96, 160
    lib = L.load()
    f01 = (torch.randn(2, H, W, device=DEV) * 9).contiguous()
    f10 = (torch.randn(2, H, W, device=DEV) * 9).contiguous()
    t = torch.tensor([0.625], device=DEV)
    outs = []
    for _ in range(3):
        acc = torch.zeros(lib.demfi_cfr_workspace_bytes(H, W) // 8, dtype=torch.int64, device=DEV)
        out = torch.zeros(4, H, W, device=DEV)
        L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), t.data_ptr(), H, W, acc.data_ptr(), out.data_ptr(),
                                         None, _stream()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    a, b = O.cfr_flow_align(f01.cpu()[None], f10.cpu()[None], t.cpu().view(1, 1, 1, 1))
    assert (outs[0] - torch.cat([a[0], b[0]], 0)).abs().max() < 1e-3
    # a workspace left dirty by an aborted launch poisons later calls; demfi_cfr_reset repairs it
    acc = torch.full((lib.demfi_cfr_workspace_bytes(H, W) // 8,), 12345, dtype=torch.int64, device=DEV)
    L.check(lib.demfi_cfr_reset(acc.data_ptr(), H, W, _stream()))
    out = torch.zeros(4, H, W, device=DEV)
    L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), t.data_ptr(), H, W, acc.data_ptr(), out.data_ptr(), None, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), outs[0]) and int(acc.abs().max()) == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_fgac_gather_and_gate(golden_dir, dtype):
    g = _gold(golden_dir, 'fgac_16x24')
    H, W, Cc = 16, 24, 64
    lib = L.load()
    ref = torch.from_numpy(g['ref']).to(DEV)
    src = ref.permute(1, 2, 0).contiguous().to(dtype)
    for name in ('inrange', 'mixed', 'beyond'):
        fl = torch.from_numpy(g['flow_' + name]).to(DEV)
        out = torch.zeros(H, W, Cc, device=DEV, dtype=dtype)
        dbg = torch.zeros(3, H * W, dtype=torch.int32, device=DEV)
        vs, vo = _view_nhwc(src), _view_nhwc(out)
        L.check(lib.demfi_fgac_gather(C.byref(vs), fl.data_ptr(), C.byref(vo), Cc, H, W, dbg.data_ptr(), _stream()))
        torch.cuda.synchronize()
        exp, m = O.fgac_sample_explicit(src.permute(2, 0, 1).float().cpu()[None], fl.cpu()[None])
        tol = 2e-6 if dtype == torch.float32 else 1e-3
        assert (out.permute(2, 0, 1).float().cpu() - exp[0]).abs().max() < tol
        inb = sum((m['inb'][k].astype(np.int32) << k) for k in range(4))
        got = dbg.cpu().numpy().reshape(3, H, W)
        assert np.array_equal(got[2] & 15, inb)
    w = torch.rand(H, W, device=DEV)
    e = torch.randn(H, W, Cc, device=DEV).to(dtype)
    o = torch.zeros(H, W, Cc, device=DEV, dtype=dtype)
    vs, ve, vo = _view_nhwc(src), _view_nhwc(e), _view_nhwc(o)
    L.check(lib.demfi_gate_blend(w.data_ptr(), C.byref(vs), C.byref(ve), C.byref(vo), Cc, H, W, _stream()))
    torch.cuda.synchronize()
    exp = w[..., None] * src.float() + (1 - w[..., None]) * e.float()
    assert (o.float() - exp).abs().max() < (1e-6 if dtype == torch.float32 else 4e-3)


def test_s2d_reflect_overlay(golden_dir):
    lib = L.load()
    H, W = 16, 24
    x = torch.randn(3, 4, H, W, device=DEV)
    for dtype, dt in ((torch.float32, L.F32), (torch.float16, L.F16)):
        out = torch.zeros(H // 2, W // 2, 48, device=DEV, dtype=dtype)
        L.check(lib.demfi_space_to_depth(x.data_ptr(), out.data_ptr(), dt, H, W, _stream()))
        torch.cuda.synchronize()
        cat = x.permute(1, 0, 2, 3).reshape(1, 12, H, W).cpu()
        exp = O.space_to_depth(cat, 2)[0].permute(1, 2, 0)
        assert torch.equal(out.cpu().float(), exp.to(dtype).float())
    ov = torch.zeros(3, H, W, device=DEV)
    L.check(lib.demfi_overlay_mean(x.data_ptr(), ov.data_ptr(), H, W, _stream()))
    xs = torch.randn(12, 50, 70, device=DEV)
    xp = torch.zeros(12, 64, 96, device=DEV)
    L.check(lib.demfi_reflect_pad(xs.data_ptr(), xp.data_ptr(), 12, 50, 70, 64, 96, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(ov.cpu(), torch.mean(x[:, 0:2], dim=1).cpu())
    assert torch.equal(xp.cpu(), torch.nn.functional.pad(xs.cpu()[None], [0, 26, 0, 14], mode='reflect')[0])
    assert lib.demfi_reflect_pad(xs.data_ptr(), xp.data_ptr(), 12, 50, 70, 128, 96, _stream()) == -1   # pad >= size


@pytest.mark.parametrize('dtype,dt', [(torch.float16, L.F16), (torch.float32, L.F32)])
def test_warp_blend_pack_writes_the_next_layers_record(dtype, dt):
    """demfi_warp_blend_pack = demfi_warp_blend on 3-channel planar frames + the NHWC record [out | fa | fb | sigmoid(logit)]
    the next convolution reads (replaces a demfi_pack_planes launch): outputs identical to the plain call, record == planes."""
    torch.manual_seed(2)
    H, W = 37, 75
    lib = L.load()
    A = torch.rand(3, H, W, device=DEV) * 2 - 1
    B = torch.rand(3, H, W, device=DEV) * 2 - 1
    fl = (torch.randn(4, H, W, device=DEV) * 5).contiguous()
    logit = torch.randn(H, W, device=DEV)
    t = torch.tensor([0.625], device=DEV)
    o1, o2 = torch.zeros(3, H, W, device=DEV), torch.zeros(3, H, W, device=DEV)
    occ1, occ2 = torch.zeros(H, W, device=DEV), torch.zeros(H, W, device=DEV)
    pk = torch.zeros(H, W, 8, device=DEV, dtype=dtype)
    va, vb, v1, v2 = _view_planar(A), _view_planar(B), _view_planar(o1), _view_planar(o2)
    L.check(lib.demfi_warp_blend(C.byref(va), fl.data_ptr(), C.byref(vb), fl[2:].data_ptr(), logit.data_ptr(), t.data_ptr(),
                                 C.byref(v1), 3, H, W, occ1.data_ptr(), None, _stream()))
    L.check(lib.demfi_warp_blend_pack(C.byref(va), fl.data_ptr(), C.byref(vb), fl[2:].data_ptr(), logit.data_ptr(), t.data_ptr(),
                                      C.byref(v2), H, W, occ2.data_ptr(), pk.data_ptr(), dt, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(occ1, occ2)
    exp = torch.cat([o1, fl, occ1[None]], 0).permute(1, 2, 0).to(dtype)
    assert torch.equal(pk, exp)
    fat = torch.zeros(H, W, 64, device=DEV, dtype=torch.float16)
    vf = _view_nhwc(fat)
    assert lib.demfi_warp_blend_pack(C.byref(vf), fl.data_ptr(), C.byref(vf), fl[2:].data_ptr(), logit.data_ptr(), t.data_ptr(),
                                     C.byref(vf), H, W, None, pk.data_ptr(), dt, _stream()) == -1
