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
