"""Window scheduler: runs x M interpolation of one 4-frame window on one GPU.

Counterpart of the inner loop of test_custom (/root/reference/main.py:1121-1178) without the PNG codec: the
unpadded window is reflect-padded straight into the engine's input buffer, the t-independent trunk runs once,
then the per-t segment runs for t = k/M.  The per-t launch sequence (~130 launches) is captured once into a
hipGraph through the C ABI and replayed (t lives in device memory, so one graph serves every t)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from .harness import t_schedule


class WindowRunner:
    def __init__(self, model, height, width, n_tst=3, mfi=8, use_graph=True):
        self.h, self.w = height, width
        H = (height + 31) // 32 * 32
        W = (width + 31) // 32 * 32
        # two per-t contexts: consecutive time instants run concurrently on two streams (they only share the trunk's
        # outputs), which fills the launch gaps and the tails of the ~107 kernels of a per-t pass
        self.n_ctx = int(os.environ.get("DEMFI_NCTX", 3)) if (use_graph and mfi > 2) else 1
        self.engine = model.engine(H, W, n_tst, n_ctx=self.n_ctx)
        self.n_tst, self.mfi = n_tst, mfi
        self.ts = [float(t) for t in t_schedule(mfi)]
        dev = self.engine.device
        self.stream = torch.cuda.Stream(dev)
        self.t_streams = [self.stream] + [torch.cuda.Stream(dev) for _ in range(self.n_ctx - 1)]
        self.out = torch.zeros((mfi - 1, 3, height, width), dtype=torch.float32, device=dev)   # St per t
        self.s01 = torch.zeros((2, 3, height, width), dtype=torch.float32, device=dev)          # S0, S1 (first t)
        self.t_all = torch.tensor(self.ts, dtype=torch.float32, device=dev)
        self.lib = L.load()
        self.use_graph = use_graph
        self._g_trunk = None
        self._g_t = None

    def _capture(self, fn, stream=None):
        h = (stream or self.stream).cuda_stream
        L.check(self.lib.demfi_graph_begin(h), 'graph_begin')
        try:
            fn(h)
        finally:
            g = C.c_void_p()
            L.check(self.lib.demfi_graph_end(h, C.byref(g)), 'graph_end')
        return g

    def _prepare_graphs(self, h):
        e = self.engine
        if self._g_trunk is not None:
            return
        e.run_trunk(h)                              # warm (module load, attributes) before capture
        for c in range(self.n_ctx):
            e.use_ctx(c)
            e.run_t(h, self.n_tst)
        self.stream.synchronize()
        self._g_trunk = self._capture(e.run_trunk)
        self._g_t = []
        for c in range(self.n_ctx):
            e.use_ctx(c)
            self._g_t.append(self._capture(lambda s: e.run_t(s, self.n_tst), self.t_streams[c]))
        e.use_ctx(0)

    def _per_t(self, emit):
        """Run the M-1 time instants, alternating over the per-t contexts / streams; emit(k, finals, stream_handle) copies
        the outputs of instant k out of the context's buffers (on that context's stream)."""
        e = self.engine
        for s in self.t_streams[1:]:
            s.wait_stream(self.stream)              # trunk outputs ready
        for k in range(self.mfi - 1):
            c = k % self.n_ctx
            st = self.t_streams[c]
            ctx = e._ctx[c]
            with torch.cuda.stream(st):
                ctx['t_dev'].copy_(self.t_all[k:k + 1], non_blocking=True)
                if self.use_graph:
                    L.check(self.lib.demfi_graph_launch(self._g_t[c], st.cuda_stream), 'graph_launch')
                else:
                    e.use_ctx(c)
                    e.run_t(st.cuda_stream, self.n_tst)
                emit(k, ctx['finals'][self.n_tst - 1], st.cuda_stream)
        for s in self.t_streams[1:]:
            self.stream.wait_stream(s)
        e.use_ctx(0)

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
                self._prepare_graphs(h)
                L.check(self.lib.demfi_graph_launch(self._g_trunk, h), 'graph_launch')
            else:
                e.run_trunk(h)

            def emit(k, fin, sh):
                self.out[k].copy_(fin[2, :, :self.h, :self.w], non_blocking=True)
                if k == 0:
                    self.s01[0].copy_(fin[0, :, :self.h, :self.w], non_blocking=True)
                    self.s01[1].copy_(fin[1, :, :self.h, :self.w], non_blocking=True)
            self._per_t(emit)
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
            if self.use_graph:
                self._prepare_graphs(h)
                L.check(self.lib.demfi_graph_launch(self._g_trunk, h), 'graph_launch')
            else:
                e.run_trunk(h)

            def emit(k, fin, sh):
                L.check(self.lib.demfi_frame_to_u8(fin[2].data_ptr(), self._out_u8[k].data_ptr(), self.h, self.w, e.H, e.W, sh), 'to_u8')
                if k == 0:
                    for i in range(2):
                        L.check(self.lib.demfi_frame_to_u8(fin[i].data_ptr(), self._s01_u8[i].data_ptr(), self.h, self.w, e.H,
                                                           e.W, sh), 'to_u8')
            self._per_t(emit)
        cur.wait_stream(self.stream)
        return self._out_u8, self._s01_u8

    def __del__(self):
        try:
            for g in [self._g_trunk] + list(self._g_t or []):
                if g is not None:
                    self.lib.demfi_graph_destroy(g)
        except Exception:
            pass
