"""Deterministic synthetic weights keyed by state_dict name.

No checkpoint ships with the reference, so parity fixtures, GPU tests and the bench
use weights regenerated identically on any machine: each tensor comes from its own
``torch.Generator`` seeded by a CRC of (seed, key).  The scale follows the reference's
``weights_init`` (xavier-normal weights, /root/reference/utils.py:173-180, applied at main.py:176);
unlike it, biases may be given a small std so that the bias path of every kernel is exercised.
"""
import zlib
import torch
from .spec import state_dict_shapes


def _key_seed(seed, key):
    return (zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


# xavier scale drives the 3-channel frame heads far outside [-1,1] (the survey measured +-10), which
# would saturate the PSNR metric; the two frame-producing layers are therefore damped.
KEY_GAIN = {'Dec_last2.weight': 0.3, 'Dec_last2_2.weight': 0.1}


# output rows that produce flows / occlusion logits: flow_01 | flow_10 | occ_0 of FF_RDB (DeMFInet.py:247-253), the refined flow / occlusion
# residuals of the UNet (rows 0:5 of dec3, 78-84) and the per-recursion deltas of FlowOcc (860-868)
FLOW_ROWS = {'FF_RDB_Module.UPNet.2': slice(128, 133), 'Refine_Module.dec3': slice(0, 5), 'Booster_Module.flow_occ.conv2': slice(0, 5)}


def synthetic_state_dict(seed=0, hp=None, gain=1.0, bias_std=0.02, dtype=torch.float32, flow_gain=1.0):
    """flow_gain < 1 (round 6, the second weight regime of the parity fixtures): the layers' rows that emit flows and occlusion logits
    are scaled down -- small motions (a few pixels instead of +-6..20) and unsaturated occlusion maps, which is what a trained
    checkpoint looks like; every other tensor is identical to flow_gain = 1."""
    sd = {}
    for key, shape in state_dict_shapes(hp).items():
        g = torch.Generator().manual_seed(_key_seed(seed, key))
        if key.endswith('.weight'):
            rf = 1
            for s in shape[2:]:
                rf *= s
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            std = gain * KEY_GAIN.get(key, 1.0) * (2.0 / (fan_in + fan_out)) ** 0.5
            sd[key] = (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
        else:
            sd[key] = (torch.randn(shape, generator=g, dtype=torch.float32) * bias_std).to(dtype)
        rows = FLOW_ROWS.get(key.rsplit('.', 1)[0])
        if rows is not None and flow_gain != 1.0:
            sd[key][rows] *= flow_gain
    return sd


def _numpy_scalar_globals():
    """What a reference checkpoint holds besides tensors: SaveManager stores the best / last metrics and loss meters next to
    ``state_dict_Model`` (/root/reference/main.py:262-271) as ``numpy.float64`` scalars, which unpickle through
    ``numpy.core.multiarray.scalar`` + ``numpy.dtype``.  Allow-listing exactly those keeps ``weights_only=True`` (no arbitrary
    code from a downloaded .pt) while genuine reference files load."""
    import numpy as np
    out = [np.dtype, np.ndarray]
    for mod in ('numpy._core.multiarray', 'numpy.core.multiarray'):
        try:
            m = __import__(mod, fromlist=['scalar'])
            out += [m.scalar, m._reconstruct]
        except (ImportError, AttributeError):
            pass
    for name in ('float64', 'float32', 'float16', 'int64', 'int32', 'uint8', 'bool_'):
        out.append(type(np.dtype(getattr(np, name))))       # numpy >= 1.25: per-type dtype classes (numpy.dtypes.Float64DType ...)
        out.append(getattr(np, name))
    seen, uniq = set(), []
    for g in out:
        if id(g) not in seen:
            seen.add(id(g))
            uniq.append(g)
    return uniq


def _safe_load(path):
    """torch.load(weights_only=True) with the numpy scalar globals of a reference checkpoint allow-listed.  Only UNPICKLING problems are
    turned into the "re-save it" advice; a missing / unreadable file raises its own OSError."""
    import pickle
    allow = _numpy_scalar_globals()
    try:
        if hasattr(torch.serialization, 'safe_globals'):            # torch >= 2.5: scoped allow-list
            with torch.serialization.safe_globals(allow):
                return torch.load(path, map_location='cpu', weights_only=True)
        if hasattr(torch.serialization, 'add_safe_globals'):        # torch 2.4: process-wide allow-list
            torch.serialization.add_safe_globals(allow)
        return torch.load(path, map_location='cpu', weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        raise ValueError('%s: cannot be read with weights_only=True (%s). Re-save it as {"state_dict_Model": model.state_dict()} '
                         'or a bare state_dict of tensors.' % (path, str(e).splitlines()[0])) from e


def load_checkpoint(path, key='state_dict_Model'):
    """state_dict of a reference checkpoint file: ``torch.load(path)['state_dict_Model']`` (/root/reference/utils.py:95-103,
    main.py:316, 351); a bare state_dict file and ``module.``-prefixed (DataParallel) keys are accepted too.  Tensors come
    back as fp32 CPU tensors, ready for ``DeMFInet.load_state_dict`` (which checks the 260 keys / shapes strictly)."""
    ck = _safe_load(path)
    sd = ck[key] if isinstance(ck, dict) and key in ck else ck
    if not isinstance(sd, dict) or not sd:
        raise ValueError('%s: no state_dict (expected a dict with %r or a bare state_dict)' % (path, key))
    out = {}
    for k, v in sd.items():
        if not torch.is_tensor(v):
            raise ValueError('%s: entry %r is not a tensor' % (path, k))
        out[k[7:] if k.startswith('module.') else k] = v.detach().to(torch.float32).cpu()
    return out


def synthetic_window(H, W, seed=1, smooth=9):
    """A 4-frame input window ``x[1,3,4,H,W]`` in [-1,1]: uniform noise low-passed by a box filter
     and re-stretched, one slowly shifted pattern per frame so that the
    network sees consistent motion."""
    g = torch.Generator().manual_seed(seed)
    pad = smooth // 2
    base = torch.rand(1, 3, H + 2 * pad + 24, W + 2 * pad + 24, generator=g) * 2 - 1
    k = torch.ones(3, 1, smooth, smooth) / (smooth * smooth)
    base = torch.nn.functional.conv2d(base, k, groups=3)
    base = base / base.abs().max()
    frames = []
    # order (B0, B1, B-1, B2): DeMFInet.py:52-55
    for shift in (8, 12, 4, 16):
        frames.append(base[:, :, shift:shift + H, 24 - shift:24 - shift + W])
    x = torch.stack(frames, 2).contiguous()
    noise = (torch.rand(x.shape, generator=g) * 2 - 1) * 0.05
    return (x + noise).clamp(-1, 1).contiguous()
