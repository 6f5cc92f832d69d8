"""GPU-box tool: where does a step of the fused residual-block kernel (resblock.hip) go?  Needs the trace build
(demfi_amd/csrc/build.sh --trace) and DEMFI_HIP_LIB=demfi_amd/csrc/libdemfi_hip_trace.so.

    DEMFI_HIP_LIB=... [PROBE_B=21] [PROBE_DATA=relu|zero] python tools/rb_trace.py

Stamps (shader cycles) per loop iteration of workgroups 0..31.  MFMA waves: 0 arrive A, 1 released, 2 conv1 MFMA phase done,
3 released from B, 4 conv1 epilogue + identity init done, 5 released from C, 6 conv2 MFMA phase done, 7 released from D, 8 conv2
epilogue done.  Helper waves: 0 window landed, 1 released from A, 2 staged outputs stored (issued), 3 released from B,
4 released from C, 5 window DMA issued, 6 released from D.  Bit 63 marks chain-opening (conv1-only) iterations."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                       # noqa: E402
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan, _Dst                  # noqa: E402

WGS, WAVES, STEPS, STAMPS = 32, 8, 24, 10
H, W = int(os.environ.get('PROBE_H', 736)), int(os.environ.get('PROBE_W', 1280))
B = int(os.environ.get('PROBE_B', 21))
DEV = 'cuda:0'


def main():
    lib = L.load()
    lib.demfi_rb_trace_dump.restype = C.c_int
    lib.demfi_rb_trace_dump.argtypes = [C.c_void_p, C.c_int64]
    pr = Plan(H, W, torch.float16, DEV)
    x, t, y = (pr._fat(H, W, 64, B) for _ in range(3))
    if os.environ.get('PROBE_DATA') == 'zero':
        x.zero_()
    elif os.environ.get('PROBE_DATA') == 'rand':
        x.copy_(torch.randn(x.shape, device=DEV) * 0.5)
    else:
        x.copy_(torch.relu(torch.randn(x.shape, device=DEV) * 0.5))
    pr.conv([], 'c1', [pr.fsrc(x, 0)], [_Dst(pr.fview(t), range(64), L.ACT_RELU)], H, W, batch=B, weight=torch.randn(64, 64, 3, 3) / 24.0, bias=torch.zeros(64))
    pr.conv([], 'c2', [pr.fsrc(t, 0)], [_Dst(pr.fview(y), range(64), L.ACT_NONE, res=pr.fview(x))], H, W, batch=B, weight=torch.randn(64, 64, 3, 3) / 24.0,
            bias=torch.zeros(64))
    pr._upload()
    st = torch.cuda.current_stream().cuda_stream
    buf = np.zeros(WGS * WAVES * STEPS * STAMPS, np.uint64)
    for _ in range(3):
        pr.launch_resblock(0, 1, st)
    torch.cuda.synchronize()
    L.check(lib.demfi_rb_trace_dump(buf.ctypes.data, buf.size))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pr.launch_resblock(0, 1, st)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    L.check(lib.demfi_rb_trace_dump(buf.ctypes.data, buf.size))
    raw = buf.reshape(WGS, WAVES, STEPS, STAMPS)
    pro = (raw[:, 0, :, 0] >> np.uint64(63)).astype(bool)                       # [wg, step]
    tr = (raw & np.uint64((1 << 63) - 1)).astype(np.int64)
    n_items = B * ((W + 29) // 30) * ((H + 15) // 16)
    print('fused residual block %dx%d batch %d: launch %.4f ms (traced build), %d items on 256 workgroups = %.1f steps each' %
          (H, W, B, ms, n_items, n_items / 256.0))
    lo, hi = 3, 22
    real = ~pro[:, lo:hi] & ~pro[:, lo + 1:hi + 1]                                # a real step followed by a real step
    m = lambda a: float(a[real].mean())
    t0 = tr[:, 0]
    period = (tr[:, 0, lo + 1:hi + 1, 1] - tr[:, 0, lo:hi, 1])
    print('  period of a step (release from A to the next one, wave 0): %.0f cycles; matrix pipe 2 x 288 x 32 = 18 432 -> busy %.2f; '
          'implied clock %.2f GHz' % (m(period), 18432.0 / m(period), m(period) * (n_items / 256.0) / (ms * 1e6)))
    for w in range(4):
        s = tr[:, w, lo:hi, :]
        nx = tr[:, w, lo + 1:hi + 1, :]
        print('  MFMA wave %d: wait A %5.0f | conv1 MFMA %6.0f | wait B %5.0f | conv1 epilogue + identity %5.0f | wait C %5.0f | conv2 MFMA %6.0f | '
              'wait D %5.0f | conv2 epilogue %5.0f | to A %5.0f' %
              (w, m(s[..., 1] - s[..., 0]), m(s[..., 2] - s[..., 1]), m(s[..., 3] - s[..., 2]), m(s[..., 4] - s[..., 3]), m(s[..., 5] - s[..., 4]),
               m(s[..., 6] - s[..., 5]), m(s[..., 7] - s[..., 6]), m(s[..., 8] - s[..., 7]), m(nx[..., 0] - s[..., 8])))
    for w in range(4, 8):
        s = tr[:, w, lo:hi, :]
        nx = tr[:, w, lo + 1:hi + 1, :]
        print('  helper %d:    wait A %5.0f | stage read + 16 stores %6.0f | wait B %6.0f | wait C %6.0f | DMA issue (19-20) %6.0f | wait D %6.0f | landing (vmcnt 0) %6.0f' %
              (w - 4, m(s[..., 1] - s[..., 0]), m(s[..., 2] - s[..., 1]), m(s[..., 3] - s[..., 2]), m(s[..., 4] - s[..., 3]), m(s[..., 5] - s[..., 4]),
               m(s[..., 6] - s[..., 5]), m(nx[..., 0] - s[..., 6])))
    op = pro[:, lo:hi]
    if op.any():
        s = tr[:, 0, lo:hi, :]
        nx = tr[:, 0, lo + 1:hi + 1, :]
        print('  chain-opening iterations seen: %d; their period %.0f cycles' % (int(op.sum()), float((nx[..., 1] - s[..., 1])[op].mean())))


if __name__ == '__main__':
    main()
