"""Seeded shape fuzz of the persistent convolution kernels on a real MI355X: the fixed cases of tests/test_gpu_kernels.py re-run at
pseudo-random frame sizes (ragged tile edges in both directions, widths that are / are not multiples of 4 -- the 16-byte vector
paths of the thin epilogue and their scalar fallbacks --, single-tile frames, more tiles than workgroups) and batch counts.  The
checks are the ones of test_gpu_kernels.py (each kernel against a torch fp64 / fp32 convolution of the same operands); the seed is
fixed, so a failure names a reproducible (case, H, W, batch)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import _lib as L                     # noqa: E402
from tests import test_gpu_kernels as K             # noqa: E402

_rng = random.Random(20260929)


def _shapes(n, hmin, hmax, wmin, wmax):
    out = []
    for i in range(n):
        H, W = _rng.randint(hmin, hmax), _rng.randint(wmin, wmax)
        if i % 3 == 0:
            W = (W + 3) & ~3                          # a third of the widths qualify for the 16-byte paths
        out.append((H, W))
    return out


C64 = [(H, W, _rng.choice([L.ACT_RELU, L.ACT_NONE]), _rng.random() < 0.6) for H, W in _shapes(10, 8, 90, 8, 200)] + [(150, 470, L.ACT_RELU, True)]


@pytest.mark.parametrize('H,W,act,res', C64)
def test_fuzz_conv3x3_64_to_64(H, W, act, res):
    """staged-store kernel (helper waves, nt output stores), with and without residual; the last case walks 304 tiles on 256 workgroups"""
    K.test_conv_vs_torch((64, 64, 3, 3, 1, H, W, act, res), torch.float16)


THIN = [(K.THIN_CASES[i % len(K.THIN_CASES)], H, W) for i, (H, W) in enumerate(_shapes(15, 8, 70, 8, 150))]


@pytest.mark.parametrize('case,H,W', THIN)
def test_fuzz_thin_outputs(case, H, W):
    K.test_narrow_persistent_conv_thin_outputs(case, H, W)


PACK = [(d, p, H, W, _rng.randint(1, 3)) for (d, p), (H, W) in
        zip([([(5, True)], [0]), ([(4, True), (1, True)], [0, 4]), ([(3, False), (5, True)], [-1, 8])] * 2, _shapes(6, 8, 60, 8, 120))]


@pytest.mark.parametrize('dsts,pack_ch,H,W,batch', PACK)
def test_fuzz_thin_outputs_with_packed_copy(dsts, pack_ch, H, W, batch):
    K.test_thin_outputs_with_packed_copy(dsts, pack_ch, H, W, batch)


NARROW = [(K.NARROW_CASES[i % len(K.NARROW_CASES)], H, W, _rng.randint(1, 2)) for i, (H, W) in enumerate(_shapes(10, 8, 70, 8, 150))]


@pytest.mark.parametrize('case,H,W,batch', NARROW)
def test_fuzz_narrow_persistent_conv(case, H, W, batch):
    K.test_narrow_persistent_conv(case, H, W, batch)


GRU = [((1, 5) if i & 1 else (5, 1), H, W, _rng.randint(1, 3)) for i, (H, W) in enumerate(_shapes(8, 8, 80, 8, 120))]


@pytest.mark.parametrize('k,H,W,batch', GRU)
def test_fuzz_sep_gru(k, H, W, batch):
    K.test_sep_gru_persistent_kernel(k[0], k[1], H, W, batch)


import random as _random                                          # its own stream: the lists above keep their round-5 shapes
_rng6 = _random.Random(606)
GRU6 = [((1, 5) if i & 1 else (5, 1), _rng6.randint(1, 20) * 8, _rng6.randint(1, 30) * 8, _rng6.randint(1, 3)) for i in range(12)]


@pytest.mark.parametrize('k,H,W,batch', GRU6)
def test_fuzz_gru_half_step_r_then_zq(k, H, W, batch):
    """Round 6 (gru.hip): seeded shapes -- strips shorter than a tile, ragged tiles along and across the filter axis, tiles whose halo
    lines are carried inside LDS and strips that restart, batch > 1."""
    K.test_gru_half_step_r_then_zq(k[0], k[1], H, W, batch)


WS = [(H, W, _rng.randint(1, 2), _rng.choice([L.ACT_TANH, L.ACT_NONE, L.ACT_RELU])) for H, W in _shapes(5, 16, 70, 32, 140)]


@pytest.mark.parametrize('H,W,batch,act', WS)
def test_fuzz_streamed_weight_conv(H, W, batch, act):
    K.test_streamed_weight_conv_7x7_192_to_64(H, W, batch, act)


GEN = [(_rng.choice([(96, 32, 3, 3), (224, 96, 1, 1), (128, 64, 4, 4), (64, 133, 3, 3), (48, 96, 5, 5), (32, 5, 3, 3)]), H, W) for H, W in _shapes(10, 1, 40, 1, 70)]


@pytest.mark.parametrize('shape,H,W', GEN)
def test_fuzz_general_kernel_small_frames(shape, H, W):
    """the general kernel at the sizes the UNet's coarse levels see (down to 1 x 1 outputs)"""
    cin, cout, kh, kw = shape
    stride = 2 if kh == 4 else 1
    K.test_conv_vs_torch((cin, cout, kh, kw, stride, H, W, L.ACT_RELU, kh == 1), torch.float16)


# ------------------------------------------------------------------------------------------------------
# whole network, fp32, against the oracle at pseudo-random frame sizes (multiples of 8: what the model itself needs), t and N
# ------------------------------------------------------------------------------------------------------
E2E = [(8 * _rng.randint(3, 14), 8 * _rng.randint(3, 22), _rng.randint(1, 3), _rng.choice([0.125, 0.3, 0.5, 0.625, 0.875]), 100 + i) for i in range(5)]


@pytest.mark.parametrize('H,W,N,tv,seed', E2E)
def test_fuzz_forward_fp32_vs_oracle(H, W, N, tv, seed):
    """DeMFInet.forward on the HIP path vs oracle.forward (= DeMFInet.py:46-179 restated) at a size none of the fixtures has: every
    Sharps_final frame within the north star's |dPSNR| <= 1e-3 dB against a common pseudo ground truth, flows / occlusion close."""
    import numpy as np
    from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
    from oracle import demfi_oracle as O
    sd = synthetic_state_dict(0)
    m = DeMFInet(HyperParams(), dtype=torch.float32)
    m.load_state_dict(sd)
    m = m.to(K.DEV).eval()
    x = synthetic_window(H, W, seed)
    t = torch.tensor([[tv]])
    got = m(x.to(K.DEV), t.to(K.DEV), N)
    with torch.no_grad():
        ref = O.forward(sd, x, t, N)
    gt = x[0, :, 1].numpy()
    for it in range(N):
        for i in range(3):
            g, r = got[1][it][i][0].cpu().numpy(), ref[1][it][i][0].numpy()
            assert np.isfinite(g).all()
            assert abs(O.psnr(g, gt) - O.psnr(r, gt)) <= 1e-3, (H, W, N, tv, it, i)
    for i in range(3):
        assert abs(O.psnr(got[0][i][0].cpu().numpy(), gt) - O.psnr(ref[0][i][0].numpy(), gt)) <= 1e-3
    assert np.median(np.abs(got[2][N][0].cpu().numpy() - ref[2][N][0].numpy())) < 2e-4
    assert np.median(np.abs(got[3][N][0].cpu().numpy() - ref[3][N][0].numpy())) < 2e-5


# ---- round 5: the fused residual block and the RDB growth convolutions (own generator: the cases above keep their shapes) --------
_rng5 = random.Random(20260930)
RESBLOCK = [(_rng5.randint(8, 120), _rng5.randint(8, 260), _rng5.randint(1, 3)) for _ in range(10)] + [(176, 900, 2)]


@pytest.mark.parametrize('H,W,batch', RESBLOCK)
def test_fuzz_fused_resblock(H, W, batch):
    """strips of 30 columns x steps of 16 rows: ragged last strip / step, chains that start mid-strip (176 x 900 x 2: 660 items on 256
    workgroups), images of fewer than 16 rows or 30 columns"""
    K.test_fused_resblock_vs_two_launches_and_torch((H, W, batch))


RDB = [(_rng5.randint(8, 200), _rng5.randint(8, 330), q) for q in (0, 1, 2, 3, 1, 2, 3, 0)]


@pytest.mark.parametrize('H,W,q', RDB)
def test_fuzz_rdb_growth_conv(H, W, q):
    """32 x 32-pixel tiles of the 3x3 streamed-weight instantiation at pseudo-random frame sizes, 3 .. 6 units of 32 channels"""
    K.test_rdb_growth_conv_streamed_weights(H, W, q)


def test_gru_first_form_of_the_zq_launch_in_its_own_process():
    """DEMFI_GRU_ZQS=0 selects the first round-6 form of the ZQ launch (z waves and q waves, q~ through LDS) -- the A/B arm of
    profiles/r06_notes.md.  The switch is read once per process, so the kernel tests run again in a child with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEMFI_GRU_ZQS='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-m', 'gpu', '-x', '-q', '-k', 'gru_half_step'],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '14 passed' in r.stdout, r.stdout[-500:]


def _even(v):
    return (v + 1) & ~1


WS2_S2 = [(K.WS2_S2_CASES[i % len(K.WS2_S2_CASES)], H, W, _rng.randint(1, 2)) for i, (H, W) in enumerate(_shapes(8, 4, 60, 4, 120))]


@pytest.mark.parametrize('case,H,W,batch', WS2_S2)
def test_fuzz_stride2_by_phases(case, H, W, batch):
    """wsconv.hip, 4x4 stride 2 as four phases of 2x2 taps: ragged 16 x 32 tiles in both directions, odd output sizes, one to four 64-cout blocks"""
    K.test_stride2_4x4_conv_by_phases(case, H, W, batch)


WS2_S1 = [(K.WS2_S1_CASES[i % len(K.WS2_S1_CASES)], _even(H), _even(W), _rng.randint(1, 2)) for i, (H, W) in enumerate(_shapes(10, 4, 60, 4, 120))]


@pytest.mark.parametrize('case,H,W,batch', WS2_S1)
def test_fuzz_conv3x3_over_units(case, H, W, batch):
    """wsconv.hip, 3x3 over 32-channel units, some read through the x2 upsample (even sizes), with and without residual"""
    K.test_conv3x3_over_32_channel_units(case, H, W, batch)


@pytest.mark.parametrize('H,W,batch', [(H, W, _rng.randint(1, 3)) for H, W in _shapes(4, 4, 50, 4, 100)])
def test_fuzz_conv3x3_units_two_piece_tail(H, W, batch):
    K.test_conv3x3_units_with_a_two_piece_tail(H, W, batch)
