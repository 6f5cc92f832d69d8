"""On-GPU evaluation of predicted frames: binding of ``demfi_eval_frame`` (PSNR + MATLAB-style SSIM of the reference's
``test()``, /root/reference/main.py:762-770, utils.py:652-705).  Frames stay in HBM; only 3 doubles per frame come back."""
import ctypes as C

import torch

from . import _lib as L


class FrameEvaluator:
    def __init__(self, h, w, device='cuda:0'):
        self.lib = L.load()
        self.h, self.w = h, w
        self.device = torch.device(device)
        self.ws = torch.zeros(self.lib.demfi_eval_workspace_bytes(h, w) // 8, dtype=torch.float64, device=self.device)

    def _planar(self, t, what):
        if not (t.is_cuda and t.device == self.device and t.dtype == torch.float32 and t.dim() == 3 and t.shape[0] == 3):
            raise ValueError('%s: fp32 [3,H,W] tensor on %s expected, got %s %s on %s' % (what, self.device, t.dtype, tuple(t.shape), t.device))
        if t.stride(2) != 1 or t.shape[1] < self.h or t.shape[2] < self.w:
            raise ValueError('%s: rows must be contiguous and at least %dx%d' % (what, self.h, self.w))
        return t.data_ptr(), t.stride(1), t.stride(0)

    def launch(self, pred, gt, out3, round_gt=False, stream=None):
        """Asynchronous: out3 = fp64[3] device tensor receiving (psnr, ssim, mse).  pred / gt may be larger (padded) buffers:
        the top-left h x w region is evaluated."""
        pp, psy, psc = self._planar(pred, 'pred')
        gp, gsy, gsc = self._planar(gt, 'gt')
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.demfi_eval_frame(pp, psy, psc, gp, gsy, gsc, self.h, self.w, 1 if round_gt else 0, self.ws.data_ptr(),
                                          out3.data_ptr(), st), 'eval_frame')

    def __call__(self, pred, gt, round_gt=False):
        out = torch.zeros(3, dtype=torch.float64, device=self.device)
        self.launch(pred, gt, out, round_gt)
        p, s, _ = out.tolist()
        return p, s


def u8_frame_to_tensor(frame_u8):
    """uint8 [h,w,3] BGR GPU tensor -> fp32 [3,h,w] in [-1,1] with the loader's arithmetic (utils.py:232-236), computed by
    the library (IEEE division: torch's GPU ``/`` multiplies by a reciprocal and differs in the last bit)."""
    if not (frame_u8.is_cuda and frame_u8.dtype == torch.uint8 and frame_u8.dim() == 3 and frame_u8.shape[2] == 3 and frame_u8.is_contiguous()):
        raise ValueError('u8_frame_to_tensor: contiguous uint8 [h,w,3] GPU tensor expected')
    h, w = frame_u8.shape[:2]
    out = torch.empty((3, h, w), dtype=torch.float32, device=frame_u8.device)
    L.check(L.load().demfi_u8_to_planar(frame_u8.data_ptr(), h, w, out.data_ptr(), torch.cuda.current_stream(frame_u8.device).cuda_stream),
            'u8_to_planar')
    return out
