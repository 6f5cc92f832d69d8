"""Frame I/O of the clip edge: counterpart of cv2.imread in the reference's loader
(/root/reference/utils.py:583-593) and cv2.imwrite in its writer (/root/reference/main.py:1165-1178).

Frames are uint8 ``[h, w, 3]`` arrays in cv2's B,G,R order.  PNG goes through the library's own zlib codec
(``demfi_png_encode`` / ``demfi_png_decode``: OpenCV / libpng headers are not in the image); ``.npy`` and raw ``.bgr``
files are the fast lane for pipelines that do not need PNG.  The ctypes calls release the GIL, so ``FramePool`` fans
frames out over host threads: 8 GPUs x ~70 frames/s is ~550 PNGs/s to encode.
"""
import collections
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib as L


def png_decode(data):
    """bytes -> uint8 [h,w,3] BGR (what ``cv2.imread(path)`` returns)."""
    lib = L.load()
    buf = np.frombuffer(data, np.uint8)
    h, w = C.c_int(0), C.c_int(0)
    L.check(lib.demfi_png_info(buf.ctypes.data, buf.size, C.byref(h), C.byref(w)), 'png_info')
    out = np.empty((h.value, w.value, 3), np.uint8)
    L.check(lib.demfi_png_decode(buf.ctypes.data, buf.size, out.ctypes.data, w.value * 3, h.value, w.value), 'png_decode')
    return out


def png_encode(img, level=1, filter=1, strategy=-1):
    """uint8 [h,w,3] BGR -> PNG bytes (8-bit RGB, lossless: decodes to the pixels ``cv2.imwrite`` would store).  Defaults =
    OpenCV's writer: zlib level 1, Sub filter, Z_RLE; filter=-1 / level 6 trade time for size."""
    lib = L.load()
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError('png_encode: uint8 [h,w,3] expected, got %s %s' % (img.dtype, img.shape))
    h, w = img.shape[:2]
    cap = lib.demfi_png_encode_bound(h, w)
    out = np.empty(cap, np.uint8)
    n = C.c_int64(0)
    L.check(lib.demfi_png_encode(img.ctypes.data, h, w, w * 3, level, filter, strategy, out.ctypes.data, cap, C.byref(n)), 'png_encode')
    return out[:n.value].tobytes()


def read_frame(path):
    """``cv2.imread`` of the reference's loader for .png; .npy / .bgr (raw, needs ``shape`` in the name: x_720x1280.bgr)."""
    if path.endswith('.npy'):
        a = np.load(path)
    elif path.endswith('.bgr'):
        hw = os.path.basename(path)[:-4].rsplit('_', 1)[-1].split('x')
        a = np.fromfile(path, np.uint8).reshape(int(hw[0]), int(hw[1]), 3)
    else:
        with open(path, 'rb') as f:
            a = png_decode(f.read())
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError('%s: uint8 [h,w,3] frame expected, got %s %s' % (path, a.dtype, a.shape))
    return a


def write_frame(path, img, level=1):
    """``cv2.imwrite`` of the reference's writer (.png), or .npy / raw .bgr."""
    if path.endswith('.npy'):
        np.save(path, np.ascontiguousarray(img))
    elif path.endswith('.bgr'):
        np.ascontiguousarray(img).tofile(path)
    else:
        data = png_encode(img, level)
        with open(path, 'wb') as f:
            f.write(data)


class FramePool:
    """Thread pool for decode / encode; ``submit_write`` copies the frame first (the caller's buffer may be a reused
    pinned staging area).  At most ``max_pending`` writes are queued: a producer that outruns the encoders blocks in
    ``submit_write`` on the oldest one, so host memory stays bounded whatever the clip length (each queued write holds
    one frame copy)."""

    def __init__(self, threads=None, max_pending=None):
        # DEMFI_IO_THREADS: this rank's share of the host cores (tools/run_node.sh sets it to nproc / ranks)
        self.threads = threads or int(os.environ.get('DEMFI_IO_THREADS', 0)) or min(32, os.cpu_count() or 4)
        self.pool = ThreadPoolExecutor(max_workers=self.threads)
        self.max_pending = max_pending or 4 * self.threads
        self._pending = collections.deque()

    def read_all(self, paths):
        return list(self.pool.map(read_frame, paths))

    def submit_read(self, path):
        """Future of ``read_frame(path)`` (streaming decode: ClipRunner keeps a few frames ahead of the GPU)."""
        return self.pool.submit(read_frame, path)

    def submit_write(self, path, img, level=1):
        while len(self._pending) >= self.max_pending:
            self._pending.popleft().result()             # back-pressure (and surfaces an encoder's exception early)
        self._pending.append(self.pool.submit(write_frame, path, np.array(img, copy=True), level))

    def wait(self):
        while self._pending:
            self._pending.popleft().result()

    def close(self):
        self.wait()
        self.pool.shutdown()
