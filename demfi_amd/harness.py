"""Boundary caller of the model: counterpart of the working subset of ``patch_forward_DeFInet_itr``
(/root/reference/utils.py:1339-1477, patch (1,1); the tiling branch is broken upstream):
reflect-pad bottom/right to a multiple of 32 -> one full-frame forward -> crop every returned map.
The pad runs in the HIP library (demfi_reflect_pad); crops are views.  Tensors stay on the GPU (the
reference's float64 numpy copies are host I/O and out of the hot path)."""
import torch

from . import _lib as L


def reflect_pad_to_multiple(x, multiple=32):
    """x [B,3,4,h,w] (cuda, fp32) -> [B,3,4,H,W] with H, W the next multiples of ``multiple``."""
    B, Cc, T, h, w = x.shape
    ph = (multiple - h % multiple) % multiple
    pw = (multiple - w % multiple) % multiple
    if ph == 0 and pw == 0:
        return x
    if not x.is_cuda:
        raise RuntimeError('demfi_amd.harness: GPU tensor required (HIP-only path)')
    x = x.contiguous().float()
    out = torch.empty((B, Cc, T, h + ph, w + pw), dtype=torch.float32, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.load().demfi_reflect_pad(x.data_ptr(), out.data_ptr(), B * Cc * T, h, w, h + ph, w + pw, st), 'reflect_pad')
    return out


def pad_forward_crop(model, x, t_value, num_update, multiple=32):
    """Returns the model's 5-tuple cropped to the input size (utils.py:1452-1476)."""
    h, w = x.shape[-2:]
    d1, fin, flows, occs, ov = model(reflect_pad_to_multiple(x, multiple), t_value, num_update)
    cr = lambda z: z[..., :h, :w]
    return ([cr(z) for z in d1], [[cr(z) for z in f] for f in fin], [cr(z) for z in flows], [cr(z) for z in occs], cr(ov))


def t_schedule(multiple_mfi):
    """t values of one x M window (utils.py:558): linspace(1/M, 1-1/M, M-1), float32."""
    import numpy as np
    return np.linspace(1 / multiple_mfi, 1 - 1 / multiple_mfi, multiple_mfi - 1).astype(np.float32)
