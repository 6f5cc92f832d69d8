"""Clip-level runner: the counterpart of ``test_custom`` (/root/reference/main.py:1108-1178) over the window list of
``make_2D_dataset_Custom_Test`` (/root/reference/utils.py:554-580), sharded over the GPUs of a node, and of the metric
half of ``test`` (/root/reference/main.py:756-838, 889-1103) over the test-set layout of ``make_2D_dataset_Test``
(/root/reference/utils.py:421-469).

A clip of T frames yields T-3 windows; window k uses frames (B0, B1, B-1, B2) = (k+1, k+2, k, k+3), produces M-1
interpolated frames ``<name of B0>_<suffix:03d>.png`` plus the deblurred S0 / S1 under the names of B0 / B1
(utils.py:565-577).  Windows are independent: rank r of ``world`` takes the contiguous block
``dist.shard_windows`` gives it, reads its own frames (3 halo frames re-read at a block edge), and no data-path
collective runs; per-rank counters are summed at the end (``dist.sum_over_ranks``).

Deblurred frames are written ONCE.  The reference's sequential loop writes S1 of window k and then S0 of window k+1 to
the same file (both are named after frame k+2), so what survives is: S0 of the window whose B0 is that frame, and the
S1 of the clip's LAST window.  ``deblurred_writes`` states exactly that, which makes the output independent of the
order in which encoder threads or ranks finish (round 2 submitted both and kept whichever finished last).
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


def deblurred_writes(k, n_windows):
    """(write S0, write S1) of global window k: what the reference's sequential loop leaves on disk (main.py:1165-1172:
    window k+1's S0 overwrites window k's S1)."""
    return True, k == n_windows - 1


def gt_names(frame_names, mfi, t_step_size):
    """Per window: ([St ground-truth names], S0 GT name, S1 GT name) inside the sharp folder (utils.py:446-455): the
    sharp frame of time instant j is numbered int(number of B0 + (t_step_size / M) * (j + 1)), zero-filled like B0."""
    out = []
    for b0, b1, _, _ in window_list(len(frame_names)):
        base = os.path.basename(frame_names[b0])
        stem, ext = base.rsplit('.', 1)
        st = [str(int(int(stem) + (t_step_size / mfi) * (j + 1))).zfill(len(stem)) + '.' + ext for j in range(mfi - 1)]
        out.append((st, base, os.path.basename(frame_names[b1])))
    return out


def deblur_time_indices(mfi):
    """(index of the time instant whose S0 is scored, index whose S1 is scored) as test() does it: S0 at the centre frame
    t = 0.5 for x8 (testIndex % 7 == 3, main.py:918-955) and at the only t for x2 (main.py:1000-1027); S1 from the LAST sample of
    a scene, i.e. its last window at the last t (main.py:633-645, 1053-1058).  Other M (not handled by the reference): first t."""
    return (3 if mfi == 8 else 0), mfi - 2


class EvalTable:
    """Per-time-index / per-scene averaging of test() (main.py:889-1103): column j < M-1 collects the metric of the j-th
    time instant of every window (PSNR_scene_1..7); column M-1 is the reference's ONE deblur column (PSNR_scene_8_deblur): the S0
    of every window at the reference's time instant plus the S1 of the scene's last window (``deblur_time_indices``).  A scene's
    value is the mean over its samples, the reported value the mean over scenes (PSNR_1..7, PSNR_8_deblur); ``total`` /
    ``deblur_total`` are the means over all samples (intp_PSNRs / deblur_PSNRs)."""

    def __init__(self, mfi):
        self.m1 = mfi - 1
        self.ncol = self.m1 + 1
        self.acc = {}                     # (scene, column) -> [sum_psnr, sum_ssim, n]

    def update(self, scene, j, psnr, ssim):
        a = self.acc.setdefault((scene, j), [0.0, 0.0, 0])
        a[0] += psnr
        a[1] += ssim
        a[2] += 1

    def update_deblur(self, scene, psnr, ssim):
        self.update(scene, self.m1, psnr, ssim)

    def keys_for(self, scenes):
        """The key list every rank must use for the reduction: all scenes x all columns, whatever this rank has seen."""
        return [(s, j) for s in sorted(scenes) for j in range(self.ncol)]

    def merge_vector(self, scenes):
        """Flat fp64 vector [sum_psnr, sum_ssim, n] per key of ``keys_for(scenes)`` (zeros where this rank has nothing):
        the same length and meaning on every rank, so it can be all-reduced."""
        missing = {s for s, _ in self.acc} - set(scenes)
        if missing:
            raise ValueError('EvalTable.merge_vector: scenes %s were updated but are not in the common scene list' % sorted(missing))
        return [v for k in self.keys_for(scenes) for v in self.acc.get(k, (0.0, 0.0, 0))]

    def merge_from(self, scenes, vec):
        """Replace the accumulators by a reduced ``merge_vector``."""
        keys = self.keys_for(scenes)
        vec = [float(v) for v in vec]
        if len(vec) != 3 * len(keys):
            raise ValueError('EvalTable.merge_from: %d values for %d keys' % (len(vec), len(keys)))
        self.acc = {}
        for i, k in enumerate(keys):
            if vec[3 * i + 2] > 0:
                self.acc[k] = [vec[3 * i], vec[3 * i + 1], int(round(vec[3 * i + 2]))]

    def all_reduce(self, scenes, device):
        """Sum the tables of all ranks (one small fp64 all-reduce over RCCL / gloo); every rank ends with the full table."""
        self.merge_from(scenes, D.sum_over_ranks(self.merge_vector(scenes), device).tolist())
        return self

    def summary(self):
        scenes = sorted({s for s, _ in self.acc})

        def column(j):
            ps = [self.acc[(s, j)][0] / self.acc[(s, j)][2] for s in scenes if (s, j) in self.acc]
            ss = [self.acc[(s, j)][1] / self.acc[(s, j)][2] for s in scenes if (s, j) in self.acc]
            return (float(np.mean(ps)) if ps else float('nan'), float(np.mean(ss)) if ss else float('nan'))

        def pooled(sel):
            a = [v for (s, j), v in self.acc.items() if sel(j)]
            n = sum(v[2] for v in a)
            return ((sum(v[0] for v in a) / n, sum(v[1] for v in a) / n) if n else (float('nan'),) * 2), n
        tot, n = pooled(lambda j: j < self.m1)
        dtot, dn = pooled(lambda j: j == self.m1)
        return {'per_index': [column(j) for j in range(self.m1)], 'total': tot, 'samples': n,
                'deblur': column(self.m1), 'deblur_total': dtot, 'deblur_samples': dn}


class _StreamedFrames:
    """``host_frames`` of ``WindowRunner.run_clip_u8`` for a folder run: frame i is decoded on the pool ``ahead`` frames
    before the runner asks for it and dropped once the window list has moved past it, so decode overlaps the GPU and host
    memory is O(ahead), not O(clip length) (round 2 decoded and pinned the whole shard first)."""

    def __init__(self, names, need, pool, ahead):
        self.names, self.pool, self.ahead = names, pool, ahead
        self.order = sorted(need)
        self.pos = 0
        self.fut = {}
        self.peak = 0

    def _submit_through(self, i):
        while self.pos < len(self.order) and self.order[self.pos] <= i + self.ahead:
            j = self.order[self.pos]
            self.fut[j] = self.pool.submit_read(self.names[j])
            self.pos += 1
        self.peak = max(self.peak, len(self.fut))

    def __getitem__(self, i):
        self._submit_through(i)
        for j in [j for j in self.fut if j < i - 3]:     # windows come in increasing order and read frames k .. k+3
            del self.fut[j]
        f = self.fut[i].result()
        t = torch.from_numpy(f)
        return t.pin_memory() if torch.cuda.is_available() else t    # the async H2D keeps the pinned block alive (caching host allocator)


class ClipRunner:
    """x M interpolation of whole clips on this rank's GPU: frames in (host uint8 BGR), frames out (sink or files)."""

    def __init__(self, model, height, width, n_tst=3, mfi=8, batch=4, world=1, rank=0, final_only=True, n_ctx=None, n_trunk=None, auto=False):
        from .runner import WindowRunner
        # the clip pipeline delivers the LAST recursion's frames only (like test_custom, utils.py:1430-1434), so the decoder
        # passes that only produce the earlier recursions' frames need not run: same delivered bytes (WindowRunner.final_only)
        self.runner = WindowRunner(model, height, width, n_tst, mfi, final_only=final_only, n_ctx=n_ctx, n_trunk=n_trunk, auto=auto)
        self.h, self.w, self.mfi, self.batch = height, width, mfi, batch
        self.world, self.rank = world, rank
        self.ts = t_schedule(mfi)

    def my_windows(self, n_frames):
        wins = window_list(n_frames)
        lo, hi = D.shard_windows(len(wins), self.world, self.rank)
        return lo, wins[lo:hi]

    def run_frames(self, frames, sink=None):
        """frames: sequence of uint8 [h,w,3] numpy arrays / CPU tensors of ONE clip, indexable by frame number (every rank
        passes the same list or at least its own slice populated; a lazy mapping such as the folder run's streamed decoder
        works too: each frame is fetched once, when its first window is uploaded).  sink(k, St, S0S1) is called with the
        GLOBAL window index.  Returns windows run."""
        lo, wins = self.my_windows(len(frames))

        class _Pinned:
            def __getitem__(_, i):
                f = frames[i]
                t = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
                return t if t.is_pinned() else t.pin_memory()
        shifted = (lambda k, st, s01: sink(lo + k, st, s01)) if sink is not None else None
        return self.runner.run_clip_u8(_Pinned(), wins, shifted, batch=self.batch)

    def run_folder(self, scene_dir, out_dir=None, pool=None, ext='.png', ahead=None):
        """One scene folder of PNG frames -> ``out_dir`` (default: the reference's ``<scene>_sharply_interpolated_xM``).
        Decode runs ``ahead`` frames (default: two batches) in front of the GPU on the pool's threads, encode behind it
        through the pool's bounded queue.  Returns (windows, frames written) of this rank."""
        names = sorted(os.path.join(scene_dir, f) for f in os.listdir(scene_dir) if f.endswith(ext))
        if len(names) < 4:
            raise RuntimeError('Found %d frames in %s: a clip needs at least 4' % (len(names), scene_dir))
        own = pool is None
        pool = pool or clipio.FramePool()
        lo, wins = self.my_windows(len(names))
        n_windows = len(names) - 3
        need = sorted({i for w in wins for i in w})
        frames = _StreamedFrames(names, need, pool, ahead if ahead is not None else 2 * self.batch + 3)
        out_dir = out_dir or (scene_dir.rstrip(os.sep) + '_sharply_interpolated_x' + str(self.mfi))
        os.makedirs(out_dir, exist_ok=True)
        onames = output_names(names, self.mfi)
        written = [0]

        def sink(k, st, s01):                            # k: window index inside this rank's block
            st_names, s0n, s1n = onames[lo + k]
            for j, nm in enumerate(st_names):
                pool.submit_write(os.path.join(out_dir, nm), st[j].numpy())
            w0, w1 = deblurred_writes(lo + k, n_windows)
            if w0:
                pool.submit_write(os.path.join(out_dir, s0n), s01[0].numpy())
            if w1:
                pool.submit_write(os.path.join(out_dir, s1n), s01[1].numpy())
            written[0] += len(st_names) + int(w0) + int(w1)
        n = self.runner.run_clip_u8(frames, wins, sink, batch=self.batch)
        pool.wait()
        self.last_decode_peak = frames.peak              # frames held by the streamed decoder at its fullest (tests)
        if own:
            pool.close()
        return n, written[0]

    def evaluate(self, blur_dir, sharp_dir, scene=None, t_step_size=8, tables=None, ext='.png'):
        """The metric half of test() (main.py:756-838) for one scene of the test-set layout (utils.py:421-469): blurry
        frames in ``blur_dir``, sharp ground truth in ``sharp_dir``; every window of this rank's block runs through the
        fp32-output path of the runner, and PSNR / MATLAB-SSIM of the D1 (``Sharps_prime``) and D2 (``Sharps_final[-1]``)
        frames against the ground truth are computed on the GPU (prediction rounded, target not) and accumulated per
        time index: ``tables`` = {'D1': EvalTable, 'D2': EvalTable} (created when None).  The deblur column follows the
        reference's bookkeeping (``deblur_time_indices``): S0 of every window at t = 0.5 (x8; the only t at x2), S1 of the scene's
        LAST window at the last t.  Reduce over ranks with ``EvalTable.all_reduce``.  Returns (tables, windows evaluated)."""
        from .metrics import FrameEvaluator, u8_frame_to_tensor
        names = sorted(os.path.join(blur_dir, f) for f in os.listdir(blur_dir) if f.endswith(ext))
        if len(names) < 4:
            raise RuntimeError('Found %d frames in %s: a clip needs at least 4' % (len(names), blur_dir))
        scene = scene if scene is not None else os.path.basename(blur_dir.rstrip(os.sep))
        tables = tables or {'D1': EvalTable(self.mfi), 'D2': EvalTable(self.mfi)}
        lo, wins = self.my_windows(len(names))
        n_windows = len(names) - 3
        gts = gt_names(names, self.mfi, t_step_size)
        dev = self.runner.engine.device
        ev = FrameEvaluator(self.h, self.w, dev)
        m1 = self.mfi - 1
        j_s0, j_s1 = deblur_time_indices(self.mfi)
        res = torch.zeros((2, m1 + 2, 3), dtype=torch.float64, device=dev)
        cache = {}

        def frame(path):                                 # uint8 file -> fp32 [3,h,w] in [-1,1] on the GPU (loader arithmetic)
            t = cache.get(path)
            if t is None:
                if len(cache) > 16:
                    cache.clear()
                t = cache[path] = u8_frame_to_tensor(torch.from_numpy(clipio.read_frame(path)).to(dev))
            return t
        for k, win in enumerate(wins):
            x = torch.stack([frame(names[i]) for i in win], 1).unsqueeze(0)
            st, s01, st1, s011 = self.runner.run_window(x, with_d1=True, s0_at=j_s0, s1_at=j_s1)
            st_gt, s0_gt, s1_gt = gts[lo + k]
            last = lo + k == n_windows - 1               # the scene's last sample carries the S1 score (main.py:633-645, 1053-1058)
            for j in range(m1):
                g = frame(os.path.join(sharp_dir, st_gt[j]))
                ev.launch(st1[j], g, res[0, j])
                ev.launch(st[j], g, res[1, j])
            for i, nm in enumerate((s0_gt, s1_gt)):
                if i == 1 and not last:
                    continue
                g = frame(os.path.join(sharp_dir, nm))
                ev.launch(s011[i], g, res[0, m1 + i])
                ev.launch(s01[i], g, res[1, m1 + i])
            r = res.cpu().numpy()                        # one small D2H per window; also orders the next window's reuse of the buffers
            for j in range(m1):
                tables['D1'].update(scene, j, float(r[0, j, 0]), float(r[0, j, 1]))
                tables['D2'].update(scene, j, float(r[1, j, 0]), float(r[1, j, 1]))
            for i in range(2 if last else 1):
                tables['D1'].update_deblur(scene, float(r[0, m1 + i, 0]), float(r[0, m1 + i, 1]))
                tables['D2'].update_deblur(scene, float(r[1, m1 + i, 0]), float(r[1, m1 + i, 1]))
        return tables, len(wins)

    def totals(self, windows, frames, device):
        """Sum of the per-rank counters over all ranks (the only end-of-run collective)."""
        return D.sum_over_ranks([float(windows), float(frames)], device).tolist()


def main(argv=None):
    """``python -m demfi_amd.clip <custom_path> [<out_root>] [--checkpoint PATH] ...`` -- the folder-in / folder-out command
    of ``main.py --phase test_custom`` (/root/reference/main.py:1108-1196): ``custom_path`` holds one folder of PNG frames per
    scene (make_2D_dataset_Custom_Test, utils.py:554-580); every scene is interpolated x M and written to
    ``<out_root or custom_path>/<scene>_sharply_interpolated_x<M>`` under the reference's file names.  ``--checkpoint`` loads the
    reference's ``.pt`` (``checkpoint['state_dict_Model']``, main.py:316, 351); without it the deterministic random-init weights
    are used (and said so).  Launched under ``torch.distributed.run`` the windows of every scene are sharded over the ranks
    (one GPU each, RCCL broadcast of the state_dict from rank 0, no data-path collective)."""
    import argparse
    import json
    import time
    ap = argparse.ArgumentParser(prog='python -m demfi_amd.clip', description=main.__doc__.split('\n\n')[0])
    ap.add_argument('custom_path', help='folder with one sub-folder of frames per scene (or a single scene folder of frames)')
    ap.add_argument('out_root', nargs='?', default=None, help='where the <scene>_sharply_interpolated_xM folders go (default: custom_path)')
    ap.add_argument('--checkpoint', default='', help="reference checkpoint (.pt holding 'state_dict_Model')")
    ap.add_argument('--mfi', type=int, default=8, help='multiple_MFI (main.py:98)')
    ap.add_argument('--n-tst', type=int, default=3, help='N_tst recursive boosts (main.py:101)')
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'fp32'])
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--ext', default='.png')
    ap.add_argument('--all-recursions', action='store_true', help='compute every Sharps_final entry (default: the last one only, which is all that is written)')
    ap.add_argument('--n-ctx', type=int, default=None, help='per-t contexts per launch sequence (default: fixed rule, 7 at x8); must divide M-1')
    ap.add_argument('--n-trunk', type=int, default=None, help='trunk buffer sets = windows in flight (default: 3 if they fit half of the GPU memory, else 2)')
    ap.add_argument('--auto', action='store_true', help='size n_ctx / n_trunk from the FREE memory of the GPU (shared or smaller GPUs)')
    a = ap.parse_args(argv)
    from . import DeMFInet, HyperParams, synthetic_state_dict
    from .weights import load_checkpoint
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    if not torch.cuda.is_available():
        raise SystemExit('demfi_amd.clip: no GPU visible -- the forward path is HIP-only (no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    D.init(world, rank, local)
    model = DeMFInet(HyperParams(gpu=local), dtype=torch.float16 if a.dtype == 'fp16' else torch.float32)
    if rank == 0:
        model.load_state_dict(load_checkpoint(a.checkpoint) if a.checkpoint else synthetic_state_dict(0))
    model = model.to(dev).eval()
    D.broadcast_state_dict(model, world, device=dev)
    root = a.custom_path.rstrip(os.sep)
    scenes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)) and '_sharply_interpolated_x' not in d)
    if scenes:
        jobs = [(s, os.path.join(root, s)) for s in scenes]
        out_root = a.out_root or root
    else:                                                # a single scene folder
        jobs = [(os.path.basename(root), root)]
        out_root = a.out_root or os.path.dirname(root)
    pool = clipio.FramePool()
    runners = {}
    tot_w = tot_f = 0
    t0 = time.perf_counter()
    for scene, path in jobs:
        names = sorted(f for f in os.listdir(path) if f.endswith(a.ext))
        if len(names) < 4:
            print('demfi_amd.clip: %s has %d frames (< 4), skipped' % (path, len(names)))
            continue
        h, w = clipio.read_frame(os.path.join(path, names[0])).shape[:2]
        cr = runners.get((h, w))
        if cr is None:
            cr = runners[(h, w)] = ClipRunner(model, h, w, a.n_tst, a.mfi, batch=a.batch, world=world, rank=rank,
                                              final_only=not a.all_recursions, n_ctx=a.n_ctx, n_trunk=a.n_trunk, auto=a.auto)
        nw, nf = cr.run_folder(path, os.path.join(out_root, scene + '_sharply_interpolated_x' + str(a.mfi)), pool=pool, ext=a.ext)
        tot_w += nw
        tot_f += nf
    pool.close()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    tw, tf = (D.sum_over_ranks([float(tot_w), float(tot_f)], dev).tolist() if world > 1 else (tot_w, tot_f))
    if rank == 0:
        print(json.dumps({'scenes': len(jobs), 'windows': int(tw), 'png_written': int(tf), 'seconds': round(dt, 2), 'ranks': world,
                          'St_frames_per_s': round(tw * (a.mfi - 1) / dt, 2) if dt > 0 else None,
                          'weights': os.path.basename(a.checkpoint) if a.checkpoint else 'synthetic_state_dict(0) (random init: no checkpoint given)',
                          'out_root': out_root}))
    D.finalize()


if __name__ == '__main__':
    main()
