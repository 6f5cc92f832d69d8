"""Window scheduler: runs x M interpolation of one 4-frame window on one GPU.

Counterpart of the inner loop of test_custom (/root/reference/main.py:1121-1178) without the PNG codec: the
unpadded window is reflect-padded straight into the engine's input buffer, the t-independent trunk runs once,
then the per-t segment runs for t = k/M.  The per-t launch sequence (~130 launches) is captured once into a
hipGraph through the C ABI and replayed (t lives in device memory, so one graph serves every t)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .harness import t_schedule


class WindowRunner:
    def __init__(self, model, height, width, n_tst=3, mfi=8, use_graph=True):
        self.h, self.w = height, width
        H = (height + 31) // 32 * 32
        W = (width + 31) // 32 * 32
        self.engine = model.engine(H, W, n_tst)
        self.n_tst, self.mfi = n_tst, mfi
        self.ts = [float(t) for t in t_schedule(mfi)]
        dev = self.engine.device
        self.stream = torch.cuda.Stream(dev)
        self.out = torch.zeros((mfi - 1, 3, height, width), dtype=torch.float32, device=dev)   # St per t
        self.s01 = torch.zeros((2, 3, height, width), dtype=torch.float32, device=dev)          # S0, S1 (first t)
        self.t_all = torch.tensor(self.ts, dtype=torch.float32, device=dev)
        self.lib = L.load()
        self.use_graph = use_graph
        self._g_trunk = None
        self._g_t = None

    def _capture(self, fn):
        h = self.stream.cuda_stream
        L.check(self.lib.demfi_graph_begin(h), 'graph_begin')
        try:
            fn(h)
        finally:
            g = C.c_void_p()
            L.check(self.lib.demfi_graph_end(h, C.byref(g)), 'graph_end')
        return g

    def run_window(self, x):
        """x: [1,3,4,h,w] fp32 on the GPU.  Returns (St [M-1,3,h,w], S0S1 [2,3,h,w]) -- views of reused buffers."""
        e = self.engine
        if tuple(x.shape) != (1, 3, 4, self.h, self.w):
            raise ValueError('run_window expects [1,3,4,%d,%d], got %s' % (self.h, self.w, tuple(x.shape)))
        x = x.contiguous().float()                  # the pad kernel reads raw [3,4,h,w] memory
        cur = torch.cuda.current_stream(e.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            h = self.stream.cuda_stream
            if (self.h, self.w) != (e.H, e.W):
                L.check(self.lib.demfi_reflect_pad(x.data_ptr(), e.x.data_ptr(), 12, self.h, self.w, e.H, e.W, h), 'pad')
            else:
                e.x.copy_(x[0], non_blocking=True)
            if self.use_graph:
                if self._g_trunk is None:
                    e.run_trunk(h)                      # warm (module load, attributes) before capture
                    e.run_t(h, self.n_tst)
                    self.stream.synchronize()
                    self._g_trunk = self._capture(e.run_trunk)
                    self._g_t = self._capture(lambda s: e.run_t(s, self.n_tst))
                L.check(self.lib.demfi_graph_launch(self._g_trunk, h), 'graph_launch')
            else:
                e.run_trunk(h)
            for k in range(self.mfi - 1):
                e.t_dev.copy_(self.t_all[k:k + 1], non_blocking=True)
                if self.use_graph:
                    L.check(self.lib.demfi_graph_launch(self._g_t, h), 'graph_launch')
                else:
                    e.run_t(h, self.n_tst)
                self.out[k].copy_(e.finals[self.n_tst - 1, 2, :, :self.h, :self.w], non_blocking=True)
                if k == 0:
                    self.s01[0].copy_(e.finals[self.n_tst - 1, 0, :, :self.h, :self.w], non_blocking=True)
                    self.s01[1].copy_(e.finals[self.n_tst - 1, 1, :, :self.h, :self.w], non_blocking=True)
        cur.wait_stream(self.stream)
        return self.out, self.s01

    def run_window_u8(self, frames_u8):
        """uint8 in / uint8 out: frames_u8 = 4 BGR uint8 [h,w,3] GPU tensors in the order (B0,B1,B-1,B2).  Returns
        (St uint8 [M-1,h,w,3], S0S1 uint8 [2,h,w,3]).  Normalisation + reflect padding are fused into ONE kernel writing
        the engine input; crop + denorm + uint8 truncation into one kernel per output frame."""
        e = self.engine
        if getattr(self, '_out_u8', None) is None:
            self._out_u8 = torch.zeros((self.mfi - 1, self.h, self.w, 3), dtype=torch.uint8, device=e.device)
            self._s01_u8 = torch.zeros((2, self.h, self.w, 3), dtype=torch.uint8, device=e.device)
        assert len(frames_u8) == 4 and all(f.dtype == torch.uint8 and tuple(f.shape) == (self.h, self.w, 3) and f.is_contiguous()
                                           for f in frames_u8)
        ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in frames_u8])
        cur = torch.cuda.current_stream(e.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            h = self.stream.cuda_stream
            L.check(self.lib.demfi_u8_to_window(ptrs, self.h, self.w, e.x.data_ptr(), e.H, e.W, h), 'u8_to_window')
            if self.use_graph and self._g_trunk is None:
                e.run_trunk(h)
                e.run_t(h, self.n_tst)
                self.stream.synchronize()
                self._g_trunk = self._capture(e.run_trunk)
                self._g_t = self._capture(lambda s: e.run_t(s, self.n_tst))
            if self.use_graph:
                L.check(self.lib.demfi_graph_launch(self._g_trunk, h), 'graph_launch')
            else:
                e.run_trunk(h)
            fin = e.finals[self.n_tst - 1]
            for k in range(self.mfi - 1):
                e.t_dev.copy_(self.t_all[k:k + 1], non_blocking=True)
                if self.use_graph:
                    L.check(self.lib.demfi_graph_launch(self._g_t, h), 'graph_launch')
                else:
                    e.run_t(h, self.n_tst)
                L.check(self.lib.demfi_frame_to_u8(fin[2].data_ptr(), self._out_u8[k].data_ptr(), self.h, self.w, e.H, e.W, h), 'to_u8')
                if k == 0:
                    for i in range(2):
                        L.check(self.lib.demfi_frame_to_u8(fin[i].data_ptr(), self._s01_u8[i].data_ptr(), self.h, self.w, e.H,
                                                           e.W, h), 'to_u8')
        cur.wait_stream(self.stream)
        return self._out_u8, self._s01_u8

    def __del__(self):
        try:
            for g in (self._g_trunk, self._g_t):
                if g is not None:
                    self.lib.demfi_graph_destroy(g)
        except Exception:
            pass
