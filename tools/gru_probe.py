"""GPU-box probe: one SepConvGRU half-step at the product shape (736x1280, batch = 7 time instants), round-5 launches against round 6.

    python tools/gru_probe.py [reps]           PROBE_B (7), PROBE_H, PROBE_W, PROBE_SETS (3: buffer sets rotated per repetition,
                                                so that the 256 MB Infinity Cache does not hold the previous repetition's tensors)
round 5:  zr (z | r as one 128-cout launch) + q          -- conv_sep5_c128_persist_kernel (conv.hip)
round 6:  r  (demfi_gru_r)  +  zq (demfi_gru_zq)        -- gru_sep5_kernel (gru.hip); also r and z through the round-5 kernel
Prints ms per launch and per half-step, TFLOP/s and the algorithmic GB/s of each launch.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan, _Dst                  # noqa: E402

H, W = int(os.environ.get('PROBE_H', 736)), int(os.environ.get('PROBE_W', 1280))
B = int(os.environ.get('PROBE_B', 7))
SETS = int(os.environ.get('PROBE_SETS', 3))
DEV = 'cuda:0'


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.manual_seed(0)
    px = H * W * B
    for kh, kw in ((1, 5), (5, 1)):
        pl = Plan(H, W, torch.float16, DEV)
        wz, wr, wq = (torch.randn(64, 128, kh, kw) * 0.04 for _ in range(3))
        bz, br, bq = (torch.randn(64) * 0.1 for _ in range(3))
        sets = []
        for s in range(SETS):
            h, x, zb, rh, hn = (pl._fat(H, W, 64, B) for _ in range(5))
            h.copy_(torch.tanh(torch.randn(h.shape, device=DEV)))
            x.copy_(torch.relu(torch.randn(x.shape, device=DEV)))          # the Mixer's output is post-ReLU
            seg = []
            i0 = len(pl._descs)
            pl.conv(seg, 'zr', [pl.fsrc(h, 0), pl.fsrc(x, 64)],
                    [_Dst(pl.fview(zb), range(0, 64), L.ACT_SIGMOID), _Dst(pl.fview(rh), range(64, 128), mode=L.MODE_MUL, res=pl.fview(h))],
                    H, W, batch=B, weight=torch.cat([wz, wr], 0), bias=torch.cat([bz, br], 0))
            pl.conv(seg, 'q', [pl.fsrc(rh, 0), pl.fsrc(x, 64)],
                    [_Dst(pl.fview(hn), range(64), mode=L.MODE_GRU, res=pl.fview(h), aux=pl.fview(zb))], H, W, batch=B, weight=wq, bias=bq)
            pl.conv(seg, 'r', [pl.fsrc(h, 0), pl.fsrc(x, 64)], [_Dst(pl.fview(rh), range(64), mode=L.MODE_MUL, res=pl.fview(h))],
                    H, W, batch=B, weight=wr, bias=br)
            pl.conv(seg, 'z', [pl.fsrc(h, 0), pl.fsrc(x, 64)], [_Dst(pl.fview(zb), range(64), L.ACT_SIGMOID)], H, W, batch=B, weight=wz, bias=bz)
            sets.append(i0)
        pl._upload()
        st = torch.cuda.current_stream().cuda_stream
        launches = {
            'r5 zr (z|r, 128 couts)': (lambda i: pl.launch_conv(i + 0, st), 2.0 * 128 * 128 * 5, 512 + 128),
            'r5 q': (lambda i: pl.launch_conv(i + 1, st), 2.0 * 64 * 128 * 5, 640),
            'r5-kernel r alone': (lambda i: pl.launch_conv(i + 2, st), 2.0 * 64 * 128 * 5, 384),
            'r6 r  (demfi_gru_r)': (lambda i: pl.launch_gru_r(i + 2, st), 2.0 * 64 * 128 * 5, 384),
            'r6 zq (demfi_gru_zq)': (lambda i: pl.launch_gru_zq(i + 3, i + 1, st), 2.0 * 128 * 128 * 5, 512),
        }
        res = {}
        for name, (fn, flop_px, bytes_px) in launches.items():
            for s in range(SETS):
                fn(sets[s])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(reps):
                fn(sets[k % SETS])
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res[name] = ms
            print('%dx%d  %-24s %7.4f ms  %7.1f TFLOP/s  %6.0f GB/s algorithmic (%d B/px)' % (kh, kw, name, ms, flop_px * px / ms / 1e9, bytes_px * px / ms / 1e6, bytes_px))
        print('%dx%d  half-step: round 5 %.4f ms   round 6 %.4f ms   (r on the round-5 kernel + zq: %.4f ms)' % (
            kh, kw, res['r5 zr (z|r, 128 couts)'] + res['r5 q'], res['r6 r  (demfi_gru_r)'] + res['r6 zq (demfi_gru_zq)'],
            res['r5-kernel r alone'] + res['r6 zq (demfi_gru_zq)']))


if __name__ == '__main__':
    main()
