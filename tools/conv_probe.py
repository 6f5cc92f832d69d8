"""GPU-box probe: time / profile single kernels at the 720p working size through the C ABI.

    python tools/conv_probe.py [case] [reps] [dtype]
cases: c3x3 (64->64 3x3 batch 3, the D1 residual conv), c3x3res, c7x7 (192->64), c1x5 (128->128 GRU zr),
       c1x1 (1152->96 half-res), warp (warp_blend fat C=64), cfr, all
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan, _Dst                  # noqa: E402

H, W = int(os.environ.get('PROBE_H', 736)), int(os.environ.get('PROBE_W', 1280))   # PROBE_H=184: tensors small enough to stay in the 256 MB Infinity Cache
B3 = int(os.environ.get('PROBE_B', 3))
DEV = 'cuda:0'


def conv_case(pl, name, cin, cout, kh, kw, batch=1, res=False, h=H, w=W, act=L.ACT_RELU):
    x = pl._fat(h, w, cin, batch)
    if os.environ.get('PROBE_DATA') == 'zero':           # power-light operands: separates clock / power effects from stalls
        x.zero_()
    elif os.environ.get('PROBE_DATA') == 'relu':         # half zeros, like post-ReLU activations
        x.copy_(torch.relu(torch.randn(x.shape, device=DEV) * 0.5))
    else:
        x.copy_(torch.randn(x.shape, device=DEV) * 0.5)
    out = pl._fat(h, w, cout, batch)
    r = pl._fat(h, w, cout, batch) if res else None
    if res:
        r.copy_(torch.randn(r.shape, device=DEV) * 0.5)
    wt = torch.randn(cout, cin, kh, kw) * (1.0 / (cin * kh * kw) ** 0.5)
    if os.environ.get('PROBE_WEIGHTS') == 'zero':
        wt.zero_()
    seg = []
    pl.conv(seg, name, [pl.fsrc(x, 0)], [_Dst(pl.fview(out), range(cout), act, res=pl.fview(r) if res else None)], h, w,
            batch=batch, weight=wt, bias=torch.zeros(cout))
    return 2.0 * cout * cin * kh * kw * h * w * batch


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'all'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == 'fp32' else torch.float16
    pl = Plan(H, W, dtype, DEV)
    cases = []
    if case in ('c3x3', 'all'):
        cases.append(('c3x3 64->64 b3', conv_case(pl, 'c3x3', 64, 64, 3, 3, B3)))
    if case in ('c3x3res', 'all'):
        cases.append(('c3x3 64->64 b3 +res', conv_case(pl, 'c3x3res', 64, 64, 3, 3, B3, True, act=L.ACT_NONE)))
    if case in ('c7x7', 'all'):
        cases.append(('c7x7 192->64', conv_case(pl, 'c7x7', 192, 64, 7, 7, act=L.ACT_TANH)))
    if case in ('c1x5', 'all'):
        cases.append(('c1x5 128->128', conv_case(pl, 'c1x5', 128, 128, 1, 5, act=L.ACT_SIGMOID)))
    if case in ('gru', 'all'):                           # the four SepConvGRU launches: zr 1x5, q 1x5, zr 5x1, q 5x1
        hb, xb, zb, rh, hn = (pl._fat(H, W, 64) for _ in range(5))
        hb.copy_(torch.tanh(torch.randn(hb.shape, device=DEV)))
        xb.copy_(torch.randn(xb.shape, device=DEV))
        for kh, kw in ((1, 5), (5, 1)):
            seg = []
            pl.conv(seg, 'zr%d%d' % (kh, kw), [pl.fsrc(hb, 0), pl.fsrc(xb, 64)],
                    [_Dst(pl.fview(zb), range(0, 64), L.ACT_SIGMOID),
                     _Dst(pl.fview(rh), range(64, 128), mode=L.MODE_MUL, res=pl.fview(hb))], H, W,
                    weight=torch.randn(128, 128, kh, kw) * 0.04, bias=torch.zeros(128))
            cases.append(('gru zr %dx%d 128->128' % (kh, kw), 2.0 * 128 * 128 * 5 * H * W))
            pl.conv(seg, 'q%d%d' % (kh, kw), [pl.fsrc(rh, 0), pl.fsrc(xb, 64)],
                    [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(hb), aux=pl.fview(zb))], H, W,
                    weight=torch.randn(64, 128, kh, kw) * 0.04, bias=torch.zeros(64))
            cases.append(('gru q %dx%d 128->64' % (kh, kw), 2.0 * 64 * 128 * 5 * H * W))
    if case in ('narrow', 'all'):                        # Mixer-shaped 3x3 layers (narrow persistent kernel)
        for nm, chs, cout in (('n8p->32', ('p8',), 32), ('n32->32', (32,), 32), ('n32->64', (32,), 64), ('n32+32->32', (32, 32), 32)):
            srcs, cin = [], 0
            for c in chs:
                if c == 'p8':
                    b = pl._fat(H, W, 8)
                    b.copy_(torch.randn(b.shape, device=DEV))
                    srcs.append(pl.fsrc_map(b, [0, 1, 2, 3, 4, -1, -1, -1]))
                    cin += 5
                else:
                    b = pl._fat(H, W, c)
                    b.copy_(torch.randn(b.shape, device=DEV))
                    srcs.append(pl.fsrc(b, cin))
                    cin += c
            o = pl._fat(H, W, cout)
            pl.conv([], nm, srcs, [_Dst(pl.fview(o), range(cout), L.ACT_RELU)], H, W, weight=torch.randn(cout, cin, 3, 3) * 0.05,
                    bias=torch.zeros(cout))
            cases.append((nm, 2.0 * cout * cin * 9 * H * W))
    if case in ('c1x1', 'all'):
        cases.append(('c1x1 1152->96 half', conv_case(pl, 'c1x1', 1152, 96, 1, 1, h=H // 2, w=W // 2, act=L.ACT_NONE)))
    if case in ('c3x3_32', 'all'):
        cases.append(('c3x3 128->32 half', conv_case(pl, 'c3x3_32', 128, 32, 3, 3, h=H // 2, w=W // 2)))
    if cases:
        pl._upload()
    st = torch.cuda.current_stream().cuda_stream
    for i, (nm, fl) in enumerate(cases):
        for _ in range(3):
            pl.launch_conv(i, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            pl.launch_conv(i, st)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-24s %8.4f ms  %7.1f TFLOP/s' % (nm, ms, fl / ms / 1e9))
    lib = L.load()
    if case in ('resblock', 'all'):
        # y = x + conv2(relu(conv1(x))): the fused kernel (round 5) against the two staged-store launches it replaces, alternating
        pr = Plan(H, W, dtype, DEV)
        x, t, y = (pr._fat(H, W, 64, B3) for _ in range(3))
        if os.environ.get('PROBE_DATA') == 'zero':
            x.zero_()
        elif os.environ.get('PROBE_DATA') == 'relu':
            x.copy_(torch.relu(torch.randn(x.shape, device=DEV) * 0.5))
        else:
            x.copy_(torch.randn(x.shape, device=DEV) * 0.5)
        pr.conv([], 'c1', [pr.fsrc(x, 0)], [_Dst(pr.fview(t), range(64), L.ACT_RELU)], H, W, batch=B3, weight=torch.randn(64, 64, 3, 3) / 24.0,
                bias=torch.zeros(64))
        pr.conv([], 'c2', [pr.fsrc(t, 0)], [_Dst(pr.fview(y), range(64), L.ACT_NONE, res=pr.fview(x))], H, W, batch=B3,
                weight=torch.randn(64, 64, 3, 3) / 24.0, bias=torch.zeros(64))
        pr._upload()
        fl = 2 * 2.0 * 64 * 64 * 9 * H * W * B3

        def timed(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps
        for rnd in range(2):
            ms2 = timed(lambda: (pr.launch_conv(0, st), pr.launch_conv(1, st)))
            ms1 = timed(lambda: pr.launch_resblock(0, 1, st))
            print('resblock b%d round %d: two launches %8.4f ms %7.1f TFLOP/s | fused %8.4f ms %7.1f TFLOP/s (%+.1f %%)'
                  % (B3, rnd, ms2, fl / ms2 / 1e9, ms1, fl / ms1 / 1e9, 100.0 * (ms1 / ms2 - 1.0)))
    if case in ('warp', 'all'):
        esz = 2 if dtype == torch.float16 else 4
        A = torch.randn(H, W, 64, device=DEV).to(dtype)
        B = torch.randn(H, W, 64, device=DEV).to(dtype)
        O = torch.zeros(H, W, 64, device=DEV, dtype=dtype)
        if os.environ.get('PROBE_SMOOTH'):          # coherent motion (what a trained network produces)
            yy, xx = torch.meshgrid(torch.arange(H, device=DEV).float(), torch.arange(W, device=DEV).float(), indexing='ij')
            fl = torch.stack([3.3 + 2 * torch.sin(yy / 37), -2.7 + 2 * torch.cos(xx / 53), -4.1 + torch.sin(xx / 41), 1.9 + torch.cos(yy / 29)]).contiguous()
        else:                                        # white-noise flows: every pixel gathers from unrelated lines
            fl = (torch.randn(4, H, W, device=DEV) * 8).contiguous()
        lg = torch.randn(H, W, device=DEV)
        t = torch.tensor([0.375], device=DEV)
        mk = lambda z: L.View(z.data_ptr(), 64, W * 64, 1, 0, 1 if dtype == torch.float32 else 0, 0)
        va, vb, vo = mk(A), mk(B), mk(O)
        run = lambda: L.check(lib.demfi_warp_blend(C.byref(va), fl.data_ptr(), C.byref(vb), fl[2:].data_ptr(), lg.data_ptr(),
                                                   t.data_ptr(), C.byref(vo), 64, H, W, None, None, st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        by = (3 * 64 * esz + 20) * H * W
        print('%-24s %8.4f ms  %7.1f GB/s (algorithmic %d B/px)' % ('warp_blend fat C=64', ms, by / ms / 1e6, 3 * 64 * esz + 20))
    if case in ('cfr', 'all'):
        f01 = (torch.randn(2, H, W, device=DEV) * 8).contiguous()
        f10 = (torch.randn(2, H, W, device=DEV) * 8).contiguous()
        acc = torch.zeros(lib.demfi_cfr_workspace_bytes(H, W) // 8, dtype=torch.int64, device=DEV)
        out = torch.zeros(4, H, W, device=DEV)
        t = torch.tensor([0.375], device=DEV)
        run = lambda: L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), t.data_ptr(), H, W, acc.data_ptr(),
                                                       out.data_ptr(), None, st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-24s %8.4f ms  %7.1f GB/s (algorithmic 32 B/px)' % ('cfr_flow_align', ms, 32.0 * H * W / ms / 1e6))


if __name__ == '__main__':
    main()
