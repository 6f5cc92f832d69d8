"""Clip-level runner: the counterpart of ``test_custom`` (/root/reference/main.py:1108-1178) over the window list of
``make_2D_dataset_Custom_Test`` (/root/reference/utils.py:554-580), sharded over the GPUs of a node.

A clip of T frames yields T-3 windows; window k uses frames (B0, B1, B-1, B2) = (k+1, k+2, k, k+3), produces M-1
interpolated frames ``<name of B0>_<suffix:03d>.png`` plus the deblurred S0 / S1 under the names of B0 / B1
(utils.py:565-577).  Windows are independent: rank r of ``world`` takes the contiguous block
``dist.shard_windows`` gives it, reads its own frames (3 halo frames re-read at a block edge), and no data-path
collective runs; per-rank counters are summed at the end (``dist.sum_over_ranks``).
"""
import os

import numpy as np
import torch

from . import clipio
from . import dist as D
from .harness import t_schedule


def window_list(n_frames):
    """(B0, B1, B-1, B2) frame indices of every window of a clip: idx = 1 .. n-3 (utils.py:564-571)."""
    return [(i, i + 1, i - 1, i + 2) for i in range(1, n_frames - 2)]


def output_names(frame_names, mfi):
    """Per window: ([St names], S0 name, S1 name) as the reference writes them (utils.py:572-577)."""
    out = []
    for b0, b1, _, _ in window_list(len(frame_names)):
        stem = os.path.basename(frame_names[b0]).split('.')[0]
        out.append(([stem + '_' + str(s).zfill(3) + '.png' for s in range(mfi - 1)],
                    os.path.basename(frame_names[b0]), os.path.basename(frame_names[b1])))
    return out


class EvalTable:
    """Per-time-index / per-scene averaging of test() (main.py:889-1103): index j collects the metric of the j-th time
    instant of every window; a scene's value is the mean over its windows, the reported value the mean over scenes;
    ``total`` is the mean over all samples (intp_PSNRs / intp_SSIMs)."""

    def __init__(self, mfi):
        self.m1 = mfi - 1
        self.acc = {}                     # (scene, j) -> [sum_psnr, sum_ssim, n]

    def update(self, scene, j, psnr, ssim):
        a = self.acc.setdefault((scene, j), [0.0, 0.0, 0])
        a[0] += psnr
        a[1] += ssim
        a[2] += 1

    def merge_vector(self):
        """Flat fp64 vector [sum_psnr, sum_ssim, n] x (scene, j) in sorted key order, for an all-reduce."""
        keys = sorted(self.acc)
        return keys, [v for k in keys for v in self.acc[k]]

    def summary(self):
        scenes = sorted({s for s, _ in self.acc})
        per_index = []
        for j in range(self.m1):
            ps = [self.acc[(s, j)][0] / self.acc[(s, j)][2] for s in scenes if (s, j) in self.acc]
            ss = [self.acc[(s, j)][1] / self.acc[(s, j)][2] for s in scenes if (s, j) in self.acc]
            per_index.append((float(np.mean(ps)) if ps else float('nan'), float(np.mean(ss)) if ss else float('nan')))
        n = sum(a[2] for a in self.acc.values())
        tot = (sum(a[0] for a in self.acc.values()) / n, sum(a[1] for a in self.acc.values()) / n) if n else (float('nan'),) * 2
        return {'per_index': per_index, 'total': tot, 'samples': n}


class ClipRunner:
    """x M interpolation of whole clips on this rank's GPU: frames in (host uint8 BGR), frames out (sink or files)."""

    def __init__(self, model, height, width, n_tst=3, mfi=8, batch=4, world=1, rank=0, final_only=True):
        from .runner import WindowRunner
        # the clip pipeline delivers the LAST recursion's frames only (like test_custom, utils.py:1430-1434), so the decoder
        # passes that only produce the earlier recursions' frames need not run: same delivered bytes (WindowRunner.final_only)
        self.runner = WindowRunner(model, height, width, n_tst, mfi, final_only=final_only)
        self.h, self.w, self.mfi, self.batch = height, width, mfi, batch
        self.world, self.rank = world, rank
        self.ts = t_schedule(mfi)

    def my_windows(self, n_frames):
        wins = window_list(n_frames)
        lo, hi = D.shard_windows(len(wins), self.world, self.rank)
        return lo, wins[lo:hi]

    def run_frames(self, frames, sink=None):
        """frames: list of uint8 [h,w,3] numpy arrays / CPU tensors of ONE clip (every rank passes the same list or at
        least its own slice populated).  sink(k, St, S0S1) is called with the GLOBAL window index.  Returns windows run."""
        lo, wins = self.my_windows(len(frames))
        need = sorted({i for w in wins for i in w})
        host = {}
        for i in need:
            f = frames[i]
            t = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
            host[i] = t if t.is_pinned() else t.pin_memory()
        shifted = (lambda k, st, s01: sink(lo + k, st, s01)) if sink is not None else None
        return self.runner.run_clip_u8(host, wins, shifted, batch=self.batch)

    def run_folder(self, scene_dir, out_dir=None, pool=None, ext='.png'):
        """One scene folder of PNG frames -> ``out_dir`` (default: the reference's ``<scene>_sharply_interpolated_xM``),
        decode / encode on a thread pool.  Returns (windows, frames written) of this rank."""
        names = sorted(os.path.join(scene_dir, f) for f in os.listdir(scene_dir) if f.endswith(ext))
        if len(names) < 4:
            raise RuntimeError('Found %d frames in %s: a clip needs at least 4' % (len(names), scene_dir))
        own = pool is None
        pool = pool or clipio.FramePool()
        lo, wins = self.my_windows(len(names))
        need = sorted({i for w in wins for i in w})
        dec = dict(zip(need, pool.read_all([names[i] for i in need])))
        frames = [dec.get(i) for i in range(len(names))]
        out_dir = out_dir or (scene_dir.rstrip(os.sep) + '_sharply_interpolated_x' + str(self.mfi))
        os.makedirs(out_dir, exist_ok=True)
        onames = output_names(names, self.mfi)
        written = [0]

        def sink(k, st, s01):
            st_names, s0n, s1n = onames[k]
            for j, nm in enumerate(st_names):
                pool.submit_write(os.path.join(out_dir, nm), st[j].numpy())
            pool.submit_write(os.path.join(out_dir, s0n), s01[0].numpy())      # main.py:1165-1172: S0 / S1 once per window
            pool.submit_write(os.path.join(out_dir, s1n), s01[1].numpy())
            written[0] += len(st_names) + 2
        n = self.run_frames(frames, sink)
        pool.wait()
        if own:
            pool.close()
        return n, written[0]

    def totals(self, windows, frames, device):
        """Sum of the per-rank counters over all ranks (the only end-of-run collective)."""
        return D.sum_over_ranks([float(windows), float(frames)], device).tolist()
