#!/usr/bin/env python
"""Headline benchmark: interpolated frames/sec @720p x8 MFI, N_tst=3 (BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one input window of a synthetic 720p clip: 4 blurry frames (resident in HBM) -> reflect-pad to
736x1280 -> t-independent trunk once -> 7 time instants t = k/8, each with N_tst = 3 recursive boosts ->
7 interpolated frames St (+ S0/S1).  Each rank owns its own windows (clip-parallel, no data-path collective;
weights are broadcast once over RCCL); value = all ranks' St frames / max-over-ranks time.  Rank 0 prints ONE
JSON line carrying the roofline of the dominant kernel and the CPU baseline (oracle, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TF = {'fp16': 2500.0, 'fp32': 157.3}      # /opt/skills/guides/MI355X_MICROARCH.md:41-42 (dense)
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md:35


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--n-tst', type=int, default=3)
    ap.add_argument('--mfi', type=int, default=8)
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--profile-ops', default='', help='write the per-launch timing table to this file')
    return ap.parse_args()


def cpu_baseline(n_tst, full_px, model=None, dev=None):
    """Oracle (CPU restatement, reference semantics: one FULL forward per t, no trunk caching) on a bounded sample:
    the full padded 736x1280 frame of the workload, ONE of the 7 t (about 30 s on 32 threads).  The same window / t is
    then pushed through the HIP path to report the metric's second half: PSNR against the fp32 reference semantics."""
    from demfi_amd import synthetic_state_dict, synthetic_window
    from oracle import demfi_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))       # beyond ~32 threads MKLDNN convs of this size slow down
    sd = synthetic_state_dict(0)
    h, w = 736, 1280
    x = synthetic_window(h, w, 1)
    t = torch.tensor([[0.5]])
    with torch.no_grad():
        O.forward(sd, synthetic_window(64, 64, 2), t, 1)          # warm-up
        t0 = time.time()
        ref = O.forward(sd, x, t, n_tst)
        dt = time.time() - t0
    frac = (h * w) / float(full_px)
    out = {'value': round(frac / dt, 5), 'unit': 'frames/s (720p-equivalent)', 'cores': torch.get_num_threads(),
           'kind': 'port', 'seconds_for_sample': round(dt, 2),
           'sample': 'oracle/demfi_oracle.forward fp32, one t, N_tst=%d, on a %dx%d window (%.3f of the padded 736x1280 '
                     'pixels); reference semantics = full forward per frame' % (n_tst, h, w, frac)}
    psnr = None
    if model is not None:
        got = model(x.to(dev), t.to(dev), n_tst)
        st = got[1][n_tst - 1][2][0].float().cpu().numpy()
        st_ref = ref[1][n_tst - 1][2][0].numpy()
        gt = x[0, :, 0].numpy()                                 # fixed pseudo ground truth
        psnr = {'St_vs_oracle_fp32_dB': round(float(O.psnr(st, st_ref)), 2),
                'dPSNR_vs_pseudoGT_dB': round(float(O.psnr(st, gt) - O.psnr(st_ref, gt)), 4),
                'note': 'same window, t=0.5, N_tst=%d; oracle = CPU fp32 restatement pinned to the reference' % n_tst}
    return out, psnr


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (a.gpus, world))
    from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
    from demfi_amd.runner import WindowRunner
    from demfi_amd import dist as D
    # test hook: DEMFI_BENCH_BACKEND=gloo runs all ranks of a multi-process launch on the GPUs that exist (rank % count),
    # so the N>1 control flow can be exercised on a 1-GPU box; the driver's 8-GPU run uses the default nccl (= RCCL)
    backend = os.environ.get('DEMFI_BENCH_BACKEND') or None
    local = local % torch.cuda.device_count() if backend == 'gloo' else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    D.init(world, rank, local, backend=backend)
    dtype = torch.float16 if a.dtype == 'fp16' else torch.float32
    model = DeMFInet(HyperParams(gpu=local), dtype=dtype)
    if rank == 0:
        model.load_state_dict(synthetic_state_dict(0))       # random-init weights of the architecture (no checkpoint offline)
    model = model.to(dev).eval()
    runner = WindowRunner(model, a.height, a.width, a.n_tst, a.mfi, use_graph=not a.no_graph)
    D.broadcast_weights(runner.engine, world)                # one flat RCCL broadcast of the repacked weights
    # synthetic clip: each rank gets its own windows (weak scaling), resident in HBM before the timed region
    nwin = a.steps + a.warmup
    windows = [synthetic_window(a.height, a.width, seed=1000 * rank + i).to(dev) for i in range(min(nwin, 4))]
    # a step = one window (trunk once + 7 time instants x N_tst boosts); the K steps are handed to the scheduler together so
    # that it can run the trunk of window w+1 under the last time instants of window w (WindowRunner.run_windows)
    out_buf = torch.empty((a.steps, a.mfi - 1, 3, a.height, a.width), dtype=torch.float32, device=dev)
    s01_buf = torch.empty((a.steps, 2, 3, a.height, a.width), dtype=torch.float32, device=dev)
    if a.warmup:
        runner.run_windows([windows[i % len(windows)] for i in range(a.warmup)])
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    runner.run_windows([windows[(a.warmup + i) % len(windows)] for i in range(a.steps)], out=out_buf, s01=s01_buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    D.barrier()
    frames = (a.mfi - 1) * a.steps * world
    eng = runner.engine
    if rank == 0:
        out = {
            # BASELINE.json's metric on its configuration; other --height/--mfi/--n-tst values label themselves
            'metric': 'interpolated frames/sec @%dp x%d MFI (N_tst=%d)' % (a.height, a.mfi, a.n_tst),
            'value': round(frames / dt, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16' if a.dtype == 'fp16' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'DeMFI-Net_rb N_tst=%d, x%d MFI, %dx%d (padded %dx%d), %s, clip-parallel windows, '
                                   'random-init weights' % (a.n_tst, a.mfi, a.height, a.width, eng.H, eng.W, a.dtype),
                       'frames_per_step': a.mfi - 1, 'graph': not a.no_graph, 'parallelism': 'clip%d' % world},
        }
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ----------
        prof = eng.profile(a.n_tst)
        per_t = sum(p[3] for p in prof if p[0] != 'trunk')
        trunk = sum(p[3] for p in prof if p[0] == 'trunk')
        convs = [p for p in prof if p[1] == 'conv']
        dom = max(convs, key=lambda p: p[3])
        grp = [p for p in convs if p[2].startswith('Decoder_res.')]          # D1 residual convs: 3x3 64->64, batch 3
        g_ms = sum(p[3] for p in grp) / len(grp)
        g_fl = 2.0 * grp[0][4]
        tot_conv_ms = sum(p[3] for p in convs)
        tot_conv_fl = 2.0 * sum(p[4] for p in convs)
        peak = MFMA_PEAK_TF[a.dtype]
        ach = g_fl / (g_ms * 1e-3) / 1e12
        out['roofline'] = {'kernel': '%s 3x3 64->64 batch 3 (D1 residual blocks, %d launches/frame)' %
                                     ('conv3x3_c64_persist_kernel<2>' if a.dtype == 'fp16' else 'conv_kernel<float,2>', len(grp)), 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                           # HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction),
                           # profiles/r01e_pmc_conv_gru_warp.txt: 738.1 MB for the 5 launches without residual (algorithmic
                           # 723.5 MB = in + out once) and 1122.0 MB for the 5 with residual (algorithmic 1085.2 MB): mean
                           'traffic': 930.1e6 if a.dtype == 'fp16' and (eng.H, eng.W) == (736, 1280) else None,
                           'algorithmic_bytes': 904.4e6 if a.dtype == 'fp16' and (eng.H, eng.W) == (736, 1280) else None,
                           'avg_launch_ms': round(g_ms, 4), 'flop_per_launch': g_fl,
                           'all_convs_TFLOPs': round(tot_conv_fl / (tot_conv_ms * 1e-3) / 1e12, 2),
                           'slowest_conv': '%s %.3f ms' % (dom[2], dom[3])}
        wb = [p for p in prof if p[1] == 'warp_fat']
        wb_ms = sum(p[3] for p in wb) / len(wb)
        esz = 2 if a.dtype == 'fp16' else 4
        wb_bytes = (3 * 64 * esz + 20) * eng.H * eng.W                       # 3*C*e + 20 B/px
        out['roofline_hbm'] = {'kernel': 'warp_blend_fat C=64 (bwarp x2 + Eq.2 blend)', 'bound': 'hbm',
                               'achieved': round(wb_bytes / (wb_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': round(wb_bytes / (wb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None,
                               'avg_launch_ms': round(wb_ms, 4), 'bytes_per_launch': wb_bytes}
        out['breakdown_ms'] = {'trunk_once_per_window': round(trunk, 2), 'per_t': round(per_t, 2),
                               'launches_per_t': len([p for p in prof if p[0] != 'trunk']),
                               'conv_share_of_per_t': round(sum(p[3] for p in convs if p[0] != 'trunk') / per_t, 3)}
        if a.profile_ops:
            with open(a.profile_ops, 'w') as f:
                for p in prof:
                    tf = 2.0 * p[4] / (p[3] * 1e-3) / 1e12 if p[4] else 0.0
                    f.write('%-7s %-10s %-48s %9.4f ms %8.1f TFLOP/s\n' % (p[0], p[1], p[2], p[3], tf))
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'], out['psnr'] = cpu_baseline(a.n_tst, eng.H * eng.W, model, dev)
        print(json.dumps(out))
    D.finalize()


if __name__ == '__main__':
    main()
