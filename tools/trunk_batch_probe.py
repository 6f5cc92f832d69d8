"""GPU-box probe: what would batching the trunks of G windows into one launch sequence buy?  Times the trunk's convolution shapes
(FF_RDB at half resolution, FAC-FB at full resolution) at batch 1 and batch G in isolation and prints per-window times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan, _Dst                  # noqa: E402

H, W, DEV = 736, 1280, 'cuda:0'
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SHAPES = [  # name, cin, cout, k, half-res?, count per trunk, base batch
    ('SFENet1 5x5 48->96', 48, 96, 5, True, 1, 1), ('SFENet2 3x3 96->96', 96, 96, 3, True, 1, 1),
    ('RDB c0 96->32', 96, 32, 3, True, 12, 1), ('RDB c1 128->32', 128, 32, 3, True, 12, 1), ('RDB c2 160->32', 160, 32, 3, True, 12, 1),
    ('RDB c3 192->32', 192, 32, 3, True, 12, 1), ('LFF 1x1 224->96', 224, 96, 1, True, 12, 1), ('GFF.0 1x1 1152->96', 1152, 96, 1, True, 1, 1),
    ('GFF.1 3x3 96->96', 96, 96, 3, True, 1, 1), ('UPNet.0 3x3 96->256', 96, 256, 3, True, 1, 1), ('UPNet.2 3x3 64->133', 64, 133, 3, False, 1, 1),
    ('w_gen 3x3 128->64 b2', 128, 64, 3, False, 1, 2), ('1x1 64->64 b2 (x2)', 64, 64, 1, False, 2, 2)]


def main():
    st = torch.cuda.current_stream().cuda_stream
    tot1 = totg = 0.0
    for name, cin, cout, k, half, cnt, b0 in SHAPES:
        h, w = (H // 2, W // 2) if half else (H, W)
        res = []
        for b in (b0, b0 * G):
            pl = Plan(h, w, torch.float16, DEV)
            x = pl._fat(h, w, cin, b)
            x.copy_(torch.relu(torch.randn(x.shape, device=DEV) * 0.5))
            o = pl._fat(h, w, cout, b)
            pl.conv([], name, [pl.fsrc(x, 0)], [_Dst(pl.fview(o), range(cout), L.ACT_RELU)], h, w, batch=b,
                    weight=torch.randn(cout, cin, k, k) * 0.02, bias=torch.zeros(cout))
            pl._upload()
            for _ in range(3):
                pl.launch_conv(0, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                pl.launch_conv(0, st)
            e1.record()
            e1.synchronize()
            res.append(e0.elapsed_time(e1) / 20)
            del pl
        t1, tg = res[0], res[1] / G
        tot1 += cnt * t1
        totg += cnt * tg
        print('%-24s x%2d  batch %d: %7.4f ms   batch %d: %7.4f ms per window (%+.0f %%)' % (name, cnt, b0, t1, b0 * G, tg, 100 * (tg / t1 - 1)))
    print('sum over the trunk\'s convolutions: %.3f ms per window at batch 1, %.3f at %d windows per launch (%+.1f %%)' % (tot1, totg, G, 100 * (totg / tot1 - 1)))


if __name__ == '__main__':
    main()
