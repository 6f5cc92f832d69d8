"""GPU-box tool: is warp_blend_fat's 0.52-0.56 of the HBM peak the kernel or the synthetic (random-weight) flows?

    python tools/warp_innet.py

In SEQUENCE, not in a repeat loop: the Ft warp of the 720p per-t plan is timed with HIP events right after
a launch that sweeps several GB through the caches (the batch-21 `Dec_first` convolution of the same plan), so F0 / F1 come from
HBM as they do in the pipeline.  Twice: with the flows the network computed (random-init weights: incoherent, +-20 px), and with
the same buffers overwritten by a smooth synthetic motion field of the same magnitude (what a trained network produces)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402

from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window    # noqa: E402
from demfi_amd.engine import SEG_TB_HEAD                 # noqa: E402
from demfi_amd.runner import WindowRunner                # noqa: E402

DEV = 'cuda:0'
H, W = 736, 1280


def main():
    m = DeMFInet(HyperParams(), dtype=torch.float16)
    m.load_state_dict(synthetic_state_dict(0))
    m = m.to(DEV).eval()
    r = WindowRunner(m, 720, 1280, n_tst=3, mfi=8, n_trunk=1)
    r.run_window(synthetic_window(720, 1280, 3).to(DEV))
    torch.cuda.synchronize()
    e = r.engine
    ops = e.ops(SEG_TB_HEAD)
    warps = [o for o in ops if o.kind == 7 and o.nch == 64][:7]          # the seven Ft warps (one per context)
    flush = [o for o in ops if o.name.decode() == 'Dec_first'][0]
    st = torch.cuda.current_stream().cuda_stream
    bytes_per_launch = (3 * 64 * 2 + 20) * H * W

    def timed(tag):
        tot, n = 0.0, 0
        for rep in range(4):
            for w in warps:
                e.run_op(flush, st)                              # 5 GB through L2 / MALL: the warp's inputs come from HBM
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e.run_op(w, st)
                e1.record()
                e1.synchronize()
                if rep:
                    tot += e0.elapsed_time(e1)
                    n += 1
        ms = tot / n
        print('%-44s %.4f ms per launch  %.2f TB/s  %.3f of 8 TB/s' % (tag, ms, bytes_per_launch / ms / 1e9, bytes_per_launch / ms / 1e9 / 8.0))

    timed('network flows (random weights, incoherent)')
    # smooth motion of the same magnitude into every context's flow_t buffer (planes: flow_t0 x, y, flow_t1 x, y)
    yy, xx = torch.meshgrid(torch.arange(H, device=DEV).float(), torch.arange(W, device=DEV).float(), indexing='ij')
    smooth = torch.stack([9.3 + 6 * torch.sin(yy / 97), -7.7 + 6 * torch.cos(xx / 131), -11.1 + 5 * torch.sin(xx / 89), 8.9 + 5 * torch.cos(yy / 73)])
    for c in range(e.n_ctx):
        ft = e._ctxs[0][c]['ft']
        mag = float(ft.abs().mean())
        ft.copy_(smooth.to(ft.dtype))
    print('mean |flow| of the network: %.1f px; smooth field: %.1f px' % (mag, float(smooth.abs().mean())))
    timed('smooth flows of comparable magnitude')


if __name__ == '__main__':
    main()
