"""CPU tests of the clip I/O edge: the PNG codec (against PIL, an independent implementation), the window list of the
reference's custom-clip loader, output naming and the evaluation table."""
import io
import os

import numpy as np
import pytest

from demfi_amd import clipio
from demfi_amd.clip import EvalTable, output_names, window_list

PIL = pytest.importorskip('PIL.Image')


def _img(h=37, w=53, seed=0):
    g = np.random.RandomState(seed)
    a = (g.rand(h, w, 3) * 255).astype(np.uint8)
    a[5:20, 8:40] = [10, 200, 30]                       # a flat patch: exercises the run-friendly filters
    a[:, :, 0] = np.minimum(a[:, :, 0], np.arange(w)[None, :] * 4 % 256)
    return a


@pytest.mark.parametrize('level,filt', [(1, -1), (0, 0), (6, 1), (3, 2), (1, 3), (9, 4)])
def test_png_roundtrip_and_pil_reads_it(level, filt):
    img = _img()
    data = clipio.png_encode(img, level, filt)
    assert np.array_equal(clipio.png_decode(data), img)
    pil = np.array(PIL.open(io.BytesIO(data)).convert('RGB'))[:, :, ::-1]          # PIL gives RGB, frames are BGR
    assert np.array_equal(pil, img)


@pytest.mark.parametrize('mode', ['RGB', 'RGBA', 'L', 'LA', 'P', 'I;16'])
def test_png_decode_of_foreign_files_matches_imread_semantics(mode):
    """cv2.imread(path) (IMREAD_COLOR): alpha dropped, gray replicated, palette expanded, 16-bit reduced to 8."""
    g = np.random.RandomState(3)
    rgb = (g.rand(24, 31, 3) * 255).astype(np.uint8)
    if mode == 'I;16':
        a16 = (g.rand(24, 31) * 65535).astype(np.uint16)
        im = PIL.fromarray(a16)                                   # 16-bit gray
        exp = np.repeat((a16 >> 8).astype(np.uint8)[:, :, None], 3, 2)
    else:
        im = PIL.fromarray(rgb).convert(mode)
        exp = np.array(im.convert('RGB'))[:, :, ::-1]
    b = io.BytesIO()
    im.save(b, 'PNG')
    assert np.array_equal(clipio.png_decode(b.getvalue()), exp)


def test_png_rejects_garbage_and_bad_crc():
    with pytest.raises(Exception):
        clipio.png_decode(b'not a png at all, definitely' * 4)
    data = bytearray(clipio.png_encode(_img()))
    data[40] ^= 0xff                                              # inside IDAT: CRC must catch it
    with pytest.raises(Exception):
        clipio.png_decode(bytes(data))


def test_frame_files_and_pool(tmp_path):
    imgs = [_img(20, 28, s) for s in range(5)]
    pool = clipio.FramePool(4)
    paths = [str(tmp_path / ('%05d.png' % i)) for i in range(5)]
    for p, im in zip(paths, imgs):
        pool.submit_write(p, im)
    pool.wait()
    back = pool.read_all(paths)
    assert all(np.array_equal(a, b) for a, b in zip(back, imgs))
    clipio.write_frame(str(tmp_path / 'a.npy'), imgs[0])
    clipio.write_frame(str(tmp_path / 'a_20x28.bgr'), imgs[1])
    assert np.array_equal(clipio.read_frame(str(tmp_path / 'a.npy')), imgs[0])
    assert np.array_equal(clipio.read_frame(str(tmp_path / 'a_20x28.bgr')), imgs[1])
    pool.close()


def test_window_list_follows_the_reference_loader():
    """make_2D_dataset_Custom_Test (utils.py:554-580): idx = 1 .. len-3, (B0,B1,B-1,B2) = (idx, idx+1, idx-1, idx+2)."""
    assert window_list(3) == [] and window_list(4) == [(1, 2, 0, 3)]
    w = window_list(11)
    assert len(w) == 8 and w[0] == (1, 2, 0, 3) and w[-1] == (8, 9, 7, 10)
    names = ['/x/scene/%05d.png' % i for i in range(6)]
    o = output_names(names, 8)
    assert len(o) == 3 and o[0][0][0] == '00001_000.png' and o[0][0][-1] == '00001_006.png'
    assert o[0][1] == '00001.png' and o[0][2] == '00002.png' and o[2][1] == '00003.png'


def test_eval_table_per_index_per_scene():
    t = EvalTable(4)
    for scene, base in (('a', 30.0), ('b', 40.0)):
        for win in range(2 if scene == 'a' else 3):
            for j in range(3):
                t.update(scene, j, base + j + win, 0.9 + 0.01 * j)
    s = t.summary()
    # index 0: scene a mean(30,31) = 30.5, scene b mean(40,41,42) = 41 -> mean over scenes 35.75
    assert abs(s['per_index'][0][0] - 35.75) < 1e-12 and abs(s['per_index'][2][1] - 0.92) < 1e-12
    assert s['samples'] == 15 and abs(s['total'][0] - np.mean([30 + j + w for w in range(2) for j in range(3)] +
                                                             [40 + j + w for w in range(3) for j in range(3)])) < 1e-12


def test_deblurred_frames_are_written_once_like_the_sequential_reference_loop():
    """main.py:1165-1172 writes S0 under B0's name and S1 under B1's name for every window, in window order: window k+1's S0
    lands on window k's S1 file.  What survives = what deblurred_writes says, so no file has two writers."""
    from demfi_amd.clip import deblurred_writes
    for n_frames in (4, 5, 9):
        wins = window_list(n_frames)
        survivor = {}
        for k, (b0, b1, _, _) in enumerate(wins):          # the reference's sequential loop
            survivor[b0] = ('S0', k)
            survivor[b1] = ('S1', k)
        ours = {}
        for k, (b0, b1, _, _) in enumerate(wins):
            w0, w1 = deblurred_writes(k, len(wins))
            if w0:
                assert b0 not in ours
                ours[b0] = ('S0', k)
            if w1:
                assert b1 not in ours
                ours[b1] = ('S1', k)
        assert ours == survivor
    # every rank decides from the GLOBAL window index: shards never write the same file
    from demfi_amd.dist import shard_windows
    n = 7
    files = []
    for rank in range(3):
        lo, hi = shard_windows(n, 3, rank)
        for k in range(lo, hi):
            w0, w1 = deblurred_writes(k, n)
            files += [k + 1] * int(w0) + [k + 2] * int(w1)
    assert sorted(files) == list(range(1, n + 2))


class _AverageClass:
    """AverageClass of the reference (utils.py:113-136): running sum / count / avg."""

    def __init__(self):
        self.sum, self.count, self.avg = 0.0, 0, 0.0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _reference_test_bookkeeping(samples, multiple):
    """The metric bookkeeping of test() restated line by line (main.py:630-700, 889-1103, x8 and x2 branches) on a list of
    per-sample values in LOADER order: (scene, intp_psnr, S0_psnr, S1_psnr), one sample per (window, t).  Returns
    (PSNR_1..M-1 scene-averaged, PSNR_8_deblur scene-averaged, intp_PSNRs.avg, deblur_PSNRs.avg)."""
    m1 = multiple - 1
    scene_cols = [_AverageClass() for _ in range(m1)]
    scene_deblur = _AverageClass()
    cols = [_AverageClass() for _ in range(m1)]
    deblur_col = _AverageClass()
    intp, deblur = _AverageClass(), _AverageClass()
    prev, last_s1 = None, None

    def close_scene():                                   # main.py:633-700 (scene change) and 1053-1103 (after the loop)
        scene_deblur.update(last_s1, 1)                  # "last sample" of each scene, S1
        deblur.update(last_s1, 1)
        for j in range(m1):
            if scene_cols[j].count:
                cols[j].update(scene_cols[j].avg, 1)
            scene_cols[j].__init__()
        deblur_col.update(scene_deblur.avg, 1)
        scene_deblur.__init__()
    for idx, (scene, p_intp, p_s0, p_s1) in enumerate(samples):
        if prev != scene and idx != 0:
            close_scene()
        last_s1 = p_s1                                   # test_psnr_S1 of the current sample (main.py:797-838)
        j = idx % m1 if multiple == 8 else 0             # main.py:889 `testIndex % (multiple - 1)`; x2: the centre frame
        scene_cols[j].update(p_intp, 1)
        intp.update(p_intp, 1)
        if (multiple == 8 and j == 3) or multiple == 2:  # main.py:918-955 / 1000-1027: S0 at t = 0.5
            scene_deblur.update(p_s0, 1)
            deblur.update(p_s0, 1)
        prev = scene
    close_scene()
    return [c.avg for c in cols], deblur_col.avg, intp.avg, deblur.avg


@pytest.mark.parametrize('mfi', [8, 2])
def test_eval_table_matches_the_bookkeeping_of_reference_test(mfi):
    """the deblur column must be the reference's: S0 of every window at t = 0.5 (x8: testIndex % 7 == 3)
    plus the S1 of each scene's LAST sample (last window, last t), one column, scene-averaged then averaged over scenes."""
    from demfi_amd.clip import deblur_time_indices
    rng = np.random.RandomState(7 + mfi)
    m1 = mfi - 1
    j_s0, j_s1 = deblur_time_indices(mfi)
    assert (j_s0, j_s1) == ((3, 6) if mfi == 8 else (0, 0))
    samples, t = [], EvalTable(mfi)
    for scene, n_win in (('sceneA', 3), ('sceneB', 1), ('sceneC', 4)):
        for k in range(n_win):
            vals = rng.uniform(20, 40, size=(m1, 3))     # per t: St, S0, S1 PSNR of this (window, t) forward
            for j in range(m1):
                samples.append((scene, vals[j, 0], vals[j, 1], vals[j, 2]))
                t.update(scene, j, vals[j, 0], vals[j, 0] / 50)
            # what ClipRunner.evaluate feeds: S0 of the window at j_s0; S1 at j_s1 for the scene's last window only
            t.update_deblur(scene, vals[j_s0, 1], vals[j_s0, 1] / 50)
            if k == n_win - 1:
                t.update_deblur(scene, vals[j_s1, 2], vals[j_s1, 2] / 50)
    cols, deb, intp, deb_all = _reference_test_bookkeeping(samples, mfi)
    s = t.summary()
    if mfi == 8:
        assert np.allclose([c[0] for c in s['per_index']], cols, rtol=0, atol=1e-12)
    else:                                                # x2 files its one column under PSNR_scene_4 (main.py:1002)
        assert abs(s['per_index'][0][0] - cols[0]) < 1e-12
    assert abs(s['deblur'][0] - deb) < 1e-12 and abs(s['total'][0] - intp) < 1e-12 and abs(s['deblur_total'][0] - deb_all) < 1e-12
    assert s['deblur_samples'] == sum(n + 1 for n in (3, 1, 4))


def test_gt_names_follow_make_2D_dataset_Test():
    """utils.py:446-455: sharp name = zfill(int(number of B0 + (t_step_size / multiple) * (mul + 1)))."""
    from demfi_amd.clip import gt_names
    names = ['/d/test_blur/s/%05d.png' % i for i in (9, 17, 25, 33, 41)]
    g = gt_names(names, 8, 8)
    assert len(g) == 2
    assert g[0][0] == ['%05d.png' % i for i in range(18, 25)] and g[0][1] == '00017.png' and g[0][2] == '00025.png'
    assert g[1][0][0] == '00026.png' and g[1][0][-1] == '00032.png'
    g2 = gt_names(['/x/%06d.png' % i for i in (1, 9, 17, 25)], 2, 8)      # x2: the centre frame only
    assert g2 == [(['000013.png'], '000009.png', '000017.png')]


def test_eval_table_reduction_vector_is_rank_consistent():
    """Ranks own different windows / scenes: the reduction vector must have the same layout everywhere (round 2 sorted each
    rank's OWN keys).  Simulated all-reduce = element-wise sum of the two ranks' vectors."""
    scenes = ['a', 'b', 'c']
    t0, t1, ref = EvalTable(4), EvalTable(4), EvalTable(4)
    for (tab, scene, j, p) in ((t0, 'a', 0, 30.0), (t0, 'a', 1, 31.0), (t1, 'a', 0, 32.0), (t1, 'c', 2, 40.0), (t1, 'c', 3, 20.0), (t0, 'b', 3, 25.0)):
        tab.update(scene, j, p, p / 100)
        ref.update(scene, j, p, p / 100)
    v0, v1 = t0.merge_vector(scenes), t1.merge_vector(scenes)
    assert len(v0) == len(v1) == 3 * len(scenes) * (3 + 1)      # M-1 interpolation columns + the ONE deblur column of test()
    t0.merge_from(scenes, [a + b for a, b in zip(v0, v1)])
    assert t0.acc == ref.acc and t0.summary() == ref.summary()
    assert t0.summary()['deblur'][0] == 22.5 and t0.summary()['deblur_samples'] == 2      # mean over scenes b (25) and c (20)
    with pytest.raises(ValueError):
        t1.merge_vector(['a'])                                  # a scene this rank updated is missing from the common list


def test_streamed_decoder_is_bounded_and_in_order(tmp_path):
    from demfi_amd.clip import _StreamedFrames
    rng = np.random.default_rng(1)
    n = 30
    names = []
    for i in range(n):
        names.append(str(tmp_path / ('%05d_12x16.bgr' % i)))
        clipio.write_frame(names[-1], rng.integers(0, 256, (12, 16, 3), dtype=np.uint8))
    pool = clipio.FramePool(3)
    sf = _StreamedFrames(names, range(n), pool, ahead=5)
    for k, win in enumerate(window_list(n)):                    # the runner asks for a window's frames in (B0,B1,B-1,B2) order
        for i in win:
            if i >= k + 2 or k == 0:                            # each frame once, when its first window is uploaded
                f = sf[i]
                assert np.array_equal(f.numpy(), clipio.read_frame(names[i]))
    assert sf.peak <= 5 + 4 + 2 and sf.pos == n                 # O(ahead) frames alive, never the whole clip
    pool.close()


def test_frame_pool_write_queue_is_bounded(tmp_path):
    pool = clipio.FramePool(2, max_pending=3)
    img = np.zeros((64, 64, 3), np.uint8)
    for i in range(20):
        pool.submit_write(str(tmp_path / ('%03d.png' % i)), img)
        assert len(pool._pending) <= 3
    pool.wait()
    assert len(os.listdir(tmp_path)) == 20
    pool.close()


def test_png_codec_rejects_crafted_headers():
    """IHDR is untrusted -- huge dimensions must come back as an error, not as std::terminate."""
    import ctypes as C
    import struct
    from demfi_amd import _lib as L
    lib = L.load()
    data = bytearray(clipio.png_encode(np.zeros((4, 4, 3), np.uint8)))
    for wv, hv in ((0x7fffffff, 0x7fffffff), (0, 4), (4, 0), (40000, 4), (20000, 20000)):
        bad = bytearray(data)
        bad[16:20] = struct.pack('>I', wv)
        bad[20:24] = struct.pack('>I', hv)
        buf = np.frombuffer(bytes(bad), np.uint8)
        h, w = C.c_int(0), C.c_int(0)
        assert lib.demfi_png_info(buf.ctypes.data, buf.size, C.byref(h), C.byref(w)) < 0
        with pytest.raises(RuntimeError):
            clipio.png_decode(bytes(bad))
    assert lib.demfi_png_encode_bound(0x7fffffff, 0x7fffffff) == 0
