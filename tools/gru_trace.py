"""GPU-box tool: where does a tile of the round-6 SepConvGRU kernel (gru.hip) go?  Needs the trace build
(demfi_amd/csrc/build.sh --trace) and DEMFI_HIP_LIB=demfi_amd/csrc/libdemfi_hip_trace.so.

    DEMFI_HIP_LIB=... [PROBE_B=7] python tools/gru_trace.py [zq|r] [1x5|5x1]

Stamps (shader cycles) per tile of workgroups 0..31.
ZQ, MFMA waves: 1 start of phase A (released from E of the previous tile), 2 phase A (h resp. r*h) done, 3 released from B, 4 phase B (x) done,
    5 released from C, 6 h prefetch + sigmoid done (z waves) / tanh done + q~ written (q waves), 7 released from D, 8 blend + staging done (z waves).
ZQ, helpers: 1 x landed, 2 next h DMA issued (after B), 3 next r*h DMA issued (after C), 4 previous stores issued + next h, r*h landed (after D),
    5 staged outputs read + next x DMA issued (after E).
R, MFMA waves: 0 arrive A, 1 released, 2 phase A (x) done, 3 released from B, 4 phase B (h) done, 5 epilogue + stores issued.
R, helpers: 0 x landed, 1 h landed, 2 x' issued, 3 h' issued."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                       # noqa: E402
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan, _Dst                  # noqa: E402

WGS, WAVES, TILES, STAMPS = 32, 8, 24, 10
H, W = int(os.environ.get('PROBE_H', 736)), int(os.environ.get('PROBE_W', 1280))
B = int(os.environ.get('PROBE_B', 7))
DEV = 'cuda:0'


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'zq'
    kh, kw = (5, 1) if len(sys.argv) > 2 and sys.argv[2] == '5x1' else (1, 5)
    lib = L.load()
    lib.demfi_gru_trace_dump.restype = C.c_int
    lib.demfi_gru_trace_dump.argtypes = [C.c_void_p, C.c_int64]
    pl = Plan(H, W, torch.float16, DEV)
    h, x, zb, rh, hn = (pl._fat(H, W, 64, B) for _ in range(5))
    h.copy_(torch.tanh(torch.randn(h.shape, device=DEV)))
    x.copy_(torch.relu(torch.randn(x.shape, device=DEV)))
    rh.copy_(h * 0.5)
    wz, wr, wq = (torch.randn(64, 128, kh, kw) * 0.04 for _ in range(3))
    bz, br, bq = (torch.randn(64) * 0.1 for _ in range(3))
    pl.conv([], 'r', [pl.fsrc(h, 0), pl.fsrc(x, 64)], [_Dst(pl.fview(rh), range(64), mode=L.MODE_MUL, res=pl.fview(h))], H, W, batch=B, weight=wr, bias=br)
    pl.conv([], 'z', [pl.fsrc(h, 0), pl.fsrc(x, 64)], [_Dst(pl.fview(zb), range(64), L.ACT_SIGMOID)], H, W, batch=B, weight=wz, bias=bz)
    pl.conv([], 'q', [pl.fsrc(rh, 0), pl.fsrc(x, 64)], [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W,
            batch=B, weight=wq, bias=bq)
    pl._upload()
    st = torch.cuda.current_stream().cuda_stream
    run = (lambda: pl.launch_gru_zq(1, 2, st)) if mode == 'zq' else (lambda: pl.launch_gru_r(0, st))
    buf = np.zeros(WGS * WAVES * TILES * STAMPS, np.uint64)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.check(lib.demfi_gru_trace_dump(buf.ctypes.data, buf.size))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    L.check(lib.demfi_gru_trace_dump(buf.ctypes.data, buf.size))
    tr = buf.reshape(WGS, WAVES, TILES, STAMPS).astype(np.int64)
    TL = 8 if mode == 'zq' else 16
    Ll, Pl = (H, W) if kh == 5 else (W, H)
    n_items = B * ((Pl + 31) // 32) * ((Ll + TL - 1) // TL)
    lo, hi = 3, 22
    m = lambda a: float(a.mean())
    period = tr[:, 0, lo + 1:hi + 1, 2] - tr[:, 0, lo:hi, 2]
    pipe = 320 * 32
    print('gru %s %dx%d  %dx%d batch %d: launch %.4f ms (traced build), %d tiles on 256 workgroups = %.1f each' % (mode, kh, kw, H, W, B, ms, n_items, n_items / 256.0))
    print('  period of a tile (release from A to the next one, wave 0): %.0f cycles; matrix pipe 320 x 32 = %d -> busy %.2f; implied clock %.2f GHz' %
          (m(period), pipe, pipe / m(period), m(period) * (n_items / 256.0) / (ms * 1e6)))
    for w in range(4):
        s = tr[:, w, lo:hi, :]
        nx = tr[:, w, lo + 1:hi + 1, :]
        d = lambda i, j: m(s[..., i] - s[..., j])
        if mode == 'zq':
            print('  MFMA wave %d (%s): phase A %6.0f | wait B %5.0f | phase B %6.0f | wait C %5.0f | %s %6.0f | wait D %5.0f | %s %6.0f | wait E %5.0f' %
                  (w, 'z' if w < 2 else 'q', d(2, 1), d(3, 2), d(4, 3), d(5, 4), 'h prefetch + sigmoid' if w < 2 else 'tanh + q~ write     ', d(6, 5), d(7, 6),
                   'blend + staging + A ring' if w < 2 else 'A ring                  ', d(8, 7), m(nx[..., 1] - s[..., 8])))
        else:
            print('  MFMA wave %d: wait A %5.0f | phase A (x) %6.0f | wait B %5.0f | phase B (h) %6.0f | epilogue + stores %6.0f | wait C + to A %5.0f' %
                  (w, d(1, 0), d(2, 1), d(3, 2), d(4, 3), d(5, 4), m(nx[..., 0] - s[..., 5])))
    for w in range(4, 8):
        s = tr[:, w, lo:hi, :]
        nx = tr[:, w, lo + 1:hi + 1, :]
        d = lambda i, j: m(s[..., i] - s[..., j])
        if mode == 'zq':
            print('  helper %d: (E) -> stage read + x\' (8) %6.0f | x landed %6.0f | (B) h\' issued (8) %6.0f | (C) rh\' issued (8) %6.0f | (D) 8 stores + h\', rh\' landed %6.0f' %
                  (w - 4, m(s[..., 5] - s[..., 4]), m(nx[..., 1] - s[..., 5]), d(2, 1), d(3, 2), d(4, 3)))
        else:
            print('  helper %d: (x landed) -> h landed %6.0f | -> x\' issued (20) %6.0f | -> h\' issued (after C) %6.0f | -> x landed %6.0f' %
                  (w - 4, d(1, 0), d(2, 1), d(3, 2), m(nx[..., 0] - s[..., 3])))


if __name__ == '__main__':
    main()
