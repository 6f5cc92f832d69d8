"""GPU-box tool: in-sequence time of named ops of the batched 720p plan (736x1280, fp16, N_tst = 3, 7 contexts).

    [DEMFI_HIP_LIB=.../libdemfi_hip_abl.so DEMFI_SEP_VARIANT=1] python tools/op_time.py <op name> [<op name> ...]

Each op is timed with HIP events right after a launch that sweeps several GB through the caches (the batch-21 `Dec_first`
convolution of the same plan), so its inputs come from HBM as they do in the pipeline; mean of 6 launches after 2 untimed ones.
With the ablation library the environment variables of conv.hip (DEMFI_SEP_VARIANT, DEMFI_PERSIST_VARIANT, ...) select a variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402

from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window    # noqa: E402
from demfi_amd.engine import SEG_TB_HEAD, SEG_TB_ITER, SEG_TRUNK                        # noqa: E402
from demfi_amd.runner import WindowRunner                # noqa: E402

DEV = 'cuda:0'


def main():
    names = sys.argv[1:]
    m = DeMFInet(HyperParams(), dtype=torch.float16)
    m.load_state_dict(synthetic_state_dict(0))
    m = m.to(DEV).eval()
    r = WindowRunner(m, 720, 1280, n_tst=3, mfi=8, n_trunk=1)
    r.run_window(synthetic_window(720, 1280, 3).to(DEV))
    torch.cuda.synchronize()
    e = r.engine
    ops = e.ops(SEG_TRUNK) + e.ops(SEG_TB_HEAD) + [o for it in range(3) for o in e.ops(SEG_TB_ITER, it=it)]
    flush = [o for o in ops if o.name.decode() == 'Dec_first'][0]
    st = torch.cuda.current_stream().cuda_stream
    tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('DEMFI_') and k != 'DEMFI_HIP_LIB')
    for name in names:
        op = [o for o in ops if o.name.decode() == name][0]
        tot, n = 0.0, 0
        for rep in range(8):
            e.run_op(flush, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e.run_op(op, st)
            e1.record()
            e1.synchronize()
            if rep >= 2:
                tot += e0.elapsed_time(e1)
                n += 1
        print('%-36s %-28s %.4f ms' % (name, tag, tot / n))


if __name__ == '__main__':
    main()
