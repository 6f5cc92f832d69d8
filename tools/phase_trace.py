"""GPU-box tool: where does a tile period of the persistent 64->64 kernels go?  Needs the trace build
(demfi_amd/csrc/build.sh --trace) and DEMFI_HIP_LIB=demfi_amd/csrc/libdemfi_hip_trace.so.

    DEMFI_HIP_LIB=... [DEMFI_PAIR=4] [PROBE_DATA=relu|zero] python tools/phase_trace.py [c3x3|c3x3res] [batch]

DEMFI_PAIR unset: the product (staged-store kernel, 4 MFMA + 4 helper waves); 4: the round-2 4-wave kernel (stores from the MFMA
waves); 1 / 2: the pair experiments (ablation build only).

The kernels stamp s_memtime (shader cycles) at their phase boundaries for the first 24 tiles of workgroups 0..31
(conv.hip: TRACE_STAMP).  Printed: mean cycles per tile of each phase, per wave role, over tiles 4..19.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                       # noqa: E402
import torch                                             # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402
from demfi_amd.engine import Plan                        # noqa: E402
import tools.conv_probe as P                             # noqa: E402

WGS, WAVES, TILES, STAMPS = 32, 10, 24, 6


def trace_plan_op(name, lib):
    """op:<name> -- one launch of a named op of the batched 720p plan (736x1280, fp16, N_tst = 3, 7 contexts), with the buffers
    a real window left behind.  Works for every persistent kernel that carries stamps (64->64 and narrow kernels)."""
    from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
    from demfi_amd.engine import SEG_TB_HEAD, SEG_TB_ITER, SEG_TRUNK
    from demfi_amd.runner import WindowRunner
    m = DeMFInet(HyperParams(), dtype=torch.float16)
    m.load_state_dict(synthetic_state_dict(0))
    m = m.to(P.DEV).eval()
    r = WindowRunner(m, 720, 1280, n_tst=3, mfi=8, n_trunk=1)
    r.run_window(synthetic_window(720, 1280, 3).to(P.DEV))
    torch.cuda.synchronize()
    e = r.engine
    ops = e.ops(SEG_TRUNK) + e.ops(SEG_TB_HEAD) + [o for it in range(3) for o in e.ops(SEG_TB_ITER, it=it)]
    op = [o for o in ops if o.name.decode() == name][0]
    st = torch.cuda.current_stream().cuda_stream
    buf = np.zeros(WGS * WAVES * TILES * STAMPS, np.uint64)
    e.run_op(op, st)
    torch.cuda.synchronize()
    L.check(lib.demfi_trace_dump(buf.ctypes.data, buf.size))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e.run_op(op, st)
    e1.record()
    e1.synchronize()
    L.check(lib.demfi_trace_dump(buf.ctypes.data, buf.size))
    tr = buf.reshape(WGS, WAVES, TILES, STAMPS).astype(np.int64)
    lo, hi = 4, 20
    print('op %s : launch %.4f ms' % (name, e0.elapsed_time(e1)))
    if name == 'Ch_Reducer':
        # streamed-weight kernel: stamp i of tile k = start of its unit i (32 input channels: 98 steps x 8 MFMAs per wave of the 4-wave kernel)
        per_tile = (tr[:, 0, hi, 0] - tr[:, 0, lo, 0]) / float(hi - lo)
        units = np.diff(tr[:, :4, lo:hi, :], axis=-1)
        tiles_per_wg = (7 * (736 // 16) * (1280 // 32)) / 256.0
        print('  cycles per tile %.0f (min %.0f max %.0f over workgroups); per unit %.0f; implied shader clock %.2f GHz (tiles per workgroup %.1f)' %
              (per_tile.mean(), per_tile.min(), per_tile.max(), units.mean(), per_tile.mean() * tiles_per_wg / (e0.elapsed_time(e1) * 1e6), tiles_per_wg))
        print('  matrix-pipe floor of a unit: 784 MFMAs x 32 cycles = 25 088 cycles per wave -> pipe busy %.2f' % (25088.0 / units.mean()))
        return
    period = (tr[:, 0, hi, 1] - tr[:, 0, lo, 1]) / float(hi - lo)
    print('  period (release to release, wave 0): mean %.0f cycles  (min %.0f max %.0f)' % (period.mean(), period.min(), period.max()))
    for w in range(WAVES):
        t, nxt = tr[:, w, lo:hi, :], tr[:, w, lo + 1:hi + 1, :]
        if not t[..., 1].any():
            continue
        if w < 4:
            print('  MFMA wave %d   wait at barrier %6.0f | MFMA phase %6.0f | epilogue %6.0f (of which waiting for the prefetched residuals / earlier stores %6.0f) | to next barrier arrival %6.0f' %
                  (w, (t[..., 1] - t[..., 0]).mean(), (t[..., 2] - t[..., 1]).mean(), (t[..., 3] - t[..., 2]).mean(),
                   (t[..., 4] - t[..., 2]).mean() if t[..., 4].any() else float('nan'), (nxt[..., 0] - t[..., 3]).mean()))
        else:
            print('  DMA wave %d    landed -> released %6.0f | issue of the next tile(s) %6.0f | issued -> landed %6.0f' %
                  (w - 4, (t[..., 1] - t[..., 0]).mean(), (t[..., 2] - t[..., 1]).mean(), (nxt[..., 0] - t[..., 2]).mean()))


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'c3x3'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = L.load()
    lib.demfi_trace_dump.restype = C.c_int
    lib.demfi_trace_dump.argtypes = [C.c_void_p, C.c_int64]
    if case.startswith('op:'):
        return trace_plan_op(case[3:], lib)
    pl = Plan(P.H, P.W, torch.float16, P.DEV)
    P.conv_case(pl, case, 64, 64, 3, 3, batch, res=(case == 'c3x3res'), act=L.ACT_NONE if case == 'c3x3res' else L.ACT_RELU)
    pl._upload()
    st = torch.cuda.current_stream().cuda_stream
    buf = np.zeros(WGS * WAVES * TILES * STAMPS, np.uint64)
    for _ in range(3):
        pl.launch_conv(0, st)
    torch.cuda.synchronize()
    L.check(lib.demfi_trace_dump(buf.ctypes.data, buf.size))        # clears
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pl.launch_conv(0, st)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    L.check(lib.demfi_trace_dump(buf.ctypes.data, buf.size))
    tr = buf.reshape(WGS, WAVES, TILES, STAMPS).astype(np.int64)
    pair = int(os.environ.get('DEMFI_PAIR', 0))
    if pair == 0:
        pair = 3                                          # the product is the staged-store kernel
    n_m = 8 if pair in (1, 2) else 4
    n_h = 4 if pair == 3 else 2                            # helper (DMA) waves
    lo, hi = 4, 20
    print('case %s batch %d  PAIR=%d  data=%s : launch %.4f ms (traced build: stamps cost a few %%)' %
          (case, batch, pair, os.environ.get('PROBE_DATA', 'rand'), ms))
    period = (tr[:, 0, hi, 1] - tr[:, 0, lo, 1]) / float(hi - lo)
    print('  period (release to release, wave 0): mean %.0f cycles  (min %.0f max %.0f over %d workgroups)' % (period.mean(), period.min(), period.max(), WGS))
    cyc = (tr[:, 0, hi, 1] - tr[:, 0, lo, 1]).mean()
    for w in range(n_m + n_h):
        t = tr[:, w, lo:hi, :]
        nxt = tr[:, w, lo + 1:hi + 1, :]
        if w < n_m:
            skew_half = pair == 1 and w >= 4
            names = ('wait at barrier', 'epilogue(k-1)' if skew_half else 'MFMA phase', 'res loads + MFMA phase' if skew_half else ('barrier B + epilogue + staging' if pair == 3 else 'epilogue'), 'to next barrier arrival')
            d = [t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], nxt[..., 0] - t[..., 3]]
            role = 'MFMA wave %d' % w
        else:
            names = ('landed -> released (wait at barrier)', 'stores + issue of next tile' if pair == 3 else 'issue of next tile', 'issued -> landed (barrier B + vmcnt wait)' if pair == 3 else 'issued -> landed (vmcnt wait)', '')
            d = [t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], nxt[..., 0] - t[..., 2], None]
            role = 'DMA wave %d' % (w - n_m)
        parts = ['%s %6.0f' % (nm, x.mean()) for nm, x in zip(names, d) if x is not None]
        if pair == 3 and w < n_m:
            parts.append('(of which wait at barrier B %6.0f)' % (t[..., 4] - t[..., 2]).mean())
        print('  %-12s %s' % (role, ' | '.join(parts)))
    # skew between waves at the barrier: who arrives last?
    arr = tr[:, :n_m + n_h, lo:hi, 0]                     # MFMA waves: arrival; DMA waves: landed
    rel = tr[:, :1, lo:hi, 1]
    late = (rel - arr).mean(axis=(0, 2))
    print('  release minus arrival per wave (the smallest one is the wave everybody waits for): ' + ' '.join('%.0f' % x for x in late))


if __name__ == '__main__':
    main()
