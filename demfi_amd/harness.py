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
    if B > 1 and x.stride(0) == 0:
        # ONE window stacked as a stride-0 batch (what a x M caller passes): pad it once and keep the stride-0 layout, which is how
        # DeMFInet.forward recognises a same-window batch (trunk once + the batched per-t plan) without reading the input
        return reflect_pad_to_multiple(x[0:1], multiple).expand(B, -1, -1, -1, -1)
    x = x.contiguous().float()
    out = torch.empty((B, Cc, T, h + ph, w + pw), dtype=torch.float32, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.load().demfi_reflect_pad(x.data_ptr(), out.data_ptr(), B * Cc * T, h, w, h + ph, w + pw, st), 'reflect_pad')
    return out


def pad_forward_crop(model, x, t_value, num_update, multiple=32, is_training=None, same_window=None):
    """Returns the model's return tuple (the 5-tuple of DeMFInet.py:178, or the 7-tuples of the visualisation / training branches)
    with every map cropped to the input size (utils.py:1452-1476).  A batch of ONE window at several t: pass it as
    ``x.expand(B, ...)`` (kept stride-0 through the pad) or say ``same_window=True`` -- the trunk then runs once."""
    h, w = x.shape[-2:]
    out = model(reflect_pad_to_multiple(x, multiple), t_value, num_update, is_training, same_window=same_window)

    def crop(z):
        if torch.is_tensor(z):
            return z[..., :h, :w]
        return [crop(y) for y in z]
    return tuple(crop(z) for z in out)


def t_schedule(multiple_mfi):
    """t values of one x M window (utils.py:558): linspace(1/M, 1-1/M, M-1), float32."""
    import numpy as np
    return np.linspace(1 / multiple_mfi, 1 - 1 / multiple_mfi, multiple_mfi - 1).astype(np.float32)


def module_window_u8(model, frames_u8, num_update, mfi):
    """The uint8 frames of one x M window computed the REFERENCE way: one full ``DeMFInet.forward`` per time instant through
    ``pad_forward_crop`` (utils.py:1339-1477), the loader's normalisation in front (utils.py:232-236) and the writer's
    ``denorm255_np`` + ``astype(uint8)`` behind (utils.py:718-721, main.py:1165-1178), each as its own kernel.
    frames_u8: 4 uint8 [h,w,3] tensors (B0, B1, B-1, B2; host or GPU).  Returns (St uint8 [M-1,h,w,3], S0S1 uint8 [2,h,w,3]) on
    the GPU -- what ``bench.py`` and the 720p test compare the scheduler's sunk bytes against, byte for byte (S0 / S1 from the
    first time instant, like main.py:1165-1172)."""
    from .metrics import u8_frame_to_tensor
    dev = model.device
    fr = [f.to(dev) for f in frames_u8]
    h, w = fr[0].shape[:2]
    x = torch.stack([u8_frame_to_tensor(f) for f in fr], 1).unsqueeze(0)          # [1,3,4,h,w]
    lib = L.load()
    st = torch.zeros((mfi - 1, h, w, 3), dtype=torch.uint8, device=dev)
    s01 = torch.zeros((2, h, w, 3), dtype=torch.uint8, device=dev)
    sh = torch.cuda.current_stream(dev).cuda_stream
    for j, t in enumerate(t_schedule(mfi)):
        fin = pad_forward_crop(model, x, torch.tensor([[float(t)]], device=dev), num_update)[1][num_update - 1]
        planes = [f[0].contiguous() for f in fin]                                  # S0, S1, St: [3,h,w] fp32
        L.check(lib.demfi_frame_to_u8(planes[2].data_ptr(), st[j].data_ptr(), h, w, h, w, sh), 'to_u8')
        if j == 0:
            for i in range(2):
                L.check(lib.demfi_frame_to_u8(planes[i].data_ptr(), s01[i].data_ptr(), h, w, h, w, sh), 'to_u8')
    torch.cuda.synchronize(dev)
    return st, s01
