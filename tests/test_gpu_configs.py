"""BASELINE.json configurations at FULL size on a real MI355X (round-2 additions):

  configs[0]  256x256, N_tst=1, x2            -> fp32 HIP forward vs the frozen reference fixture
  configs[1]  720p fp16 N_tst=3 x8            -> asserted PSNR bounds against the fp32 oracle on the full window
  configs[3]  720p fp32 N_tst=5               -> the strict |dPSNR| <= 1e-3 dB criterion at full size
  configs[4]  1080p (1088x1920) fp16 x16      -> WindowRunner(mfi=16) properties + one-t PSNR against the oracle
plus the integer index maps of the warps / splat at W = 1280 and W = 1920 (323-367 columns move in the
fp32 coordinate round trip at these widths), bit-identical to the oracle's step-by-step numpy emulation.

The oracle (CPU fp32 restatement pinned to the reference fixtures) is run ONCE per size here; the N_tst=5 run also yields
the N_tst=3 reference (the recursion is sequential: finals[2] of an N=5 forward is the N=3 result)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window   # noqa: E402
from demfi_amd import _lib as L                                                       # noqa: E402
from demfi_amd.harness import pad_forward_crop, t_schedule                            # noqa: E402
from oracle import demfi_oracle as O                                                  # noqa: E402
from tests.conftest import record_fp16_margin                                         # noqa: E402

DEV = 'cuda:0'


def _model(dtype):
    m = DeMFInet(HyperParams(), dtype=dtype)
    m.load_state_dict(synthetic_state_dict(0))
    return m.to(DEV).eval()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------------
# configs[0]
# ------------------------------------------------------------------------------------------------------
def test_config1_256_fp32_vs_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, 'cfg1_256x256_t0500_n1.npz'))
    x = synthetic_window(256, 256, int(g['seed']))
    m = _model(torch.float32)
    d1, fin, flows, occs, ov = m(x.to(DEV), torch.tensor([[0.5]], device=DEV), 1)
    st = fin[0][2][0].cpu().numpy()
    gt = x[0, :, 0].numpy()
    assert abs(O.psnr(st, gt) - float(g['psnr_St_vs_B0'])) <= 1e-3                  # north-star tolerance
    assert O.psnr(st, g['St']) > 60.0
    assert np.median(np.abs(st - g['St'])) < 2e-5
    assert np.median(np.abs(flows[-1][0].cpu().numpy() - g['flows_last'])) < 2e-4
    for i in range(3):
        d = np.abs(np.around(O.denorm255(fin[0][i][0].cpu().numpy())).astype(np.int32) - g['finals_u8'][i].astype(np.int32))
        # a 1e-5 difference flips np.around() for ~2.5e-3 of the pixels (those within 1e-5 * 127.5 of a .5 boundary)
        # (isolated floor() flips of the warps move a handful of pixels by more: DESIGN.md section 2, chaos note)
        assert (d > 0).mean() < 1e-2 and np.percentile(d, 99.9) <= 1


# ------------------------------------------------------------------------------------------------------
# configs[1] + configs[3]: one oracle run on the full 736x1280 window
# ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def oracle_720p_n5():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    x = synthetic_window(736, 1280, 1)
    t = torch.tensor([[0.5]])
    with torch.no_grad():
        ref = O.forward(synthetic_state_dict(0), x, t, 5)
    return x, t, ref


def test_config4_720p_fp32_n5_strict_psnr(oracle_720p_n5):
    """The strict-parity configuration at FULL size: fp32 path, N_tst = 5, |dPSNR| <= 1e-3 dB against a fixed pseudo
    ground truth for every frame the module returns, and the direct PSNR(build, oracle) well above 60 dB."""
    x, t, ref = oracle_720p_n5
    m = _model(torch.float32)
    d1, fin, flows, occs, ov = m(x.to(DEV), t.to(DEV), 5)
    gt = x[0, :, 0].numpy()
    for it in range(5):
        for i in range(3):
            got = fin[it][i][0].cpu().numpy()
            exp = ref[1][it][i][0].numpy()
            assert abs(O.psnr(got, gt) - O.psnr(exp, gt)) <= 1e-3, (it, i)
            assert O.psnr(got, exp) > 60.0, (it, i)
    for i in range(3):
        assert abs(O.psnr(d1[i][0].cpu().numpy(), gt) - O.psnr(ref[0][i][0].numpy(), gt)) <= 1e-3
    assert np.median(np.abs(flows[5][0].cpu().numpy() - ref[2][5][0].numpy())) < 2e-4
    del m
    torch.cuda.empty_cache()


def test_config2_720p_fp16_n3_psnr_bounds(oracle_720p_n5):
    """fp16 has no reference oracle (the reference crashes under .half()): it is stated against the fp32
    oracle.  Gate: PSNR(St, fp32) >= 44 dB and |dPSNR vs pseudo-GT| <= 5e-3 dB on the full window
    (round-1 measurement: 45.6 dB / +0.0021 dB) -- looser than the fp32 tolerance by contract, but asserted."""
    x, t, ref = oracle_720p_n5
    m = _model(torch.float16)
    d1, fin, flows, occs, ov = m(x.to(DEV), t.to(DEV), 3)
    gt = x[0, :, 0].numpy()
    for i in range(3):
        got = fin[2][i][0].cpu().numpy()
        exp = ref[1][2][i][0].numpy()
        ps, dps = O.psnr(got, exp), O.psnr(got, gt) - O.psnr(exp, gt)
        print('720p fp16 N=3 frame %d: PSNR vs fp32 oracle %.2f dB, dPSNR vs pseudo-GT %+.4f dB' % (i, ps, dps))
        record_fp16_margin('config2_720p_seed1_t0.5', i, ps, dps, size='736x1280', n_tst=3, t=0.5, seed=1)
        assert np.isfinite(got).all()
        assert ps >= 44.0 and abs(dps) <= 5e-3, (i, ps, dps)
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize('seed,tv', [(2, 0.125), (3, 0.875)])
def test_config2_720p_fp16_vs_oracle_other_windows_schedule_ends(seed, tv):
    """More DIRECT fp16-vs-oracle points at full size (the t = 0.5 test above was the only one; the
    3 x 3 gate below compares fp16 with this repo's fp32 HIP path): other windows at BOTH ends of a x8 schedule -- (seed 2, t = 1/8)
    and (seed 3, t = 7/8) -- one ~36 s oracle run at N_tst = 3 each.  Same gate; the margins go to gpurun_out/fp16_margins.json."""
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    x = synthetic_window(736, 1280, seed)
    t = torch.tensor([[tv]])
    with torch.no_grad():
        ref = O.forward(synthetic_state_dict(0), x, t, 3)
    m = _model(torch.float16)
    d1, fin, flows, occs, ov = m(x.to(DEV), t.to(DEV), 3)
    gt = x[0, :, 0].numpy()
    for i in range(3):
        got = fin[2][i][0].cpu().numpy()
        exp = ref[1][2][i][0].numpy()
        ps, dps = O.psnr(got, exp), O.psnr(got, gt) - O.psnr(exp, gt)
        print('720p fp16 N=3 seed %d t=%g frame %d: PSNR vs fp32 oracle %.2f dB (margin %.2f dB over the 44 dB gate), dPSNR vs pseudo-GT %+.4f dB'
              % (seed, tv, i, ps, ps - 44.0, dps))
        record_fp16_margin('config2_720p_seed%d_t%g' % (seed, tv), i, ps, dps, size='736x1280', n_tst=3, t=tv, seed=seed)
        assert np.isfinite(got).all()
        assert ps >= 44.0 and abs(dps) <= 5e-3, (i, ps, dps)
    del m
    torch.cuda.empty_cache()


def test_config2_720p_second_weight_regime_fp32_strict_and_fp16_margin():
    """Round 6: one full-size oracle point in the SECOND weight regime (synthetic_state_dict(flow_gain=0.3): flows of
    a few pixels, occlusion maps spread over (0, 1) -- what a trained checkpoint looks like; every other full-size test uses the
    saturated xavier regime).  fp32: |dPSNR| <= 1e-3 dB and PSNR(build, oracle) > 60 dB; fp16: the same gate as config 2, margin recorded."""
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = synthetic_state_dict(0, flow_gain=0.3)
    x = synthetic_window(736, 1280, 4)
    t = torch.tensor([[0.375]])
    with torch.no_grad():
        ref = O.forward(sd, x, t, 3)
    occ = ref[3][-1]
    print('regime 2 at 736x1280: |flow| max %.2f px, occlusion saturated (<0.02 or >0.98) on %.1f %% of the pixels' %
          (float(ref[2][-1].abs().max()), 100.0 * float(((occ < 0.02) | (occ > 0.98)).float().mean())))
    gt = x[0, :, 0].numpy()
    for dtype in (torch.float32, torch.float16):
        m = DeMFInet(HyperParams(), dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        d1, fin, flows, occs, ov = m(x.to(DEV), t.to(DEV), 3)
        for i in range(3):
            got = fin[2][i][0].cpu().numpy()
            exp = ref[1][2][i][0].numpy()
            ps, dps = O.psnr(got, exp), O.psnr(got, gt) - O.psnr(exp, gt)
            assert np.isfinite(got).all()
            if dtype == torch.float32:
                assert abs(dps) <= 1e-3 and ps > 60.0, (i, ps, dps)
            else:
                print('720p fp16 N=3 regime 2 frame %d: PSNR vs fp32 oracle %.2f dB (margin %.2f dB over the 44 dB gate), dPSNR vs pseudo-GT %+.4f dB'
                      % (i, ps, ps - 44.0, dps))
                record_fp16_margin('regime2_720p_seed4_t0.375', i, ps, dps, size='736x1280', n_tst=3, t=0.375, seed=4, flow_gain=0.3)
                assert ps >= 44.0 and abs(dps) <= 5e-3, (i, ps, dps)
        del m
        torch.cuda.empty_cache()


def test_config2_720p_runner_batched_equals_module():
    """The benched scheduler at the benched size: the 7 time instants of a 720p window as ONE launch sequence batched over 7
    per-t contexts (demfi_forward_tb; at this size the batched convolutions choose other record sizes / grids than the
    per-context ones, i.e. other packed-weight blobs) == one pad -> forward -> crop per t, bit for bit."""
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 720, 1280, 3, 8
    m = _model(torch.float16)
    x = synthetic_window(h, w, 77).to(DEV)
    runner = WindowRunner(m, h, w, n_tst=N, mfi=M)
    assert runner.tb and runner.n_ctx == 7
    st, s01 = runner.run_window(x)
    torch.cuda.synchronize()
    ts = t_schedule(M)
    for k in (0, 3, 6):
        ref = pad_forward_crop(m, x, torch.tensor([[float(ts[k])]], device=DEV), N)
        assert torch.equal(st[k], ref[1][N - 1][2][0]), k
        if k == 0:
            assert torch.equal(s01[0], ref[1][N - 1][0][0]) and torch.equal(s01[1], ref[1][N - 1][1][0])
    del runner, m
    torch.cuda.empty_cache()


def test_config2_720p_benched_path_bytes_equal_module_path():
    """the EXACT code path bench.py times -- run_clip_u8 = pinned host frames -> H2D -> fused uint8 ingest ->
    batched 7-context plan -> uint8 sink epilogue -> D2H -- at the benched size, against the reference-shaped path on the same
    frames (one DeMFInet.forward per t through pad_forward_crop, separate normalise / denorm kernels): every byte of the 7 St
    frames and of S0 / S1 of every window.  bench.py runs the same check on its last timed window and fails on a mismatch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_clip_u8
    from demfi_amd.clip import window_list
    from demfi_amd.harness import module_window_u8
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 720, 1280, 3, 8
    m = _model(torch.float16)
    frames = synthetic_clip_u8(h, w, 6, seed=4242)                     # 6 frames -> 3 windows, pinned host memory
    wins = window_list(len(frames))
    runner = WindowRunner(m, h, w, n_tst=N, mfi=M)
    assert runner.tb and runner.n_ctx == 7 and runner.engine.supports_u8_sink
    got = {}
    n = runner.run_clip_u8(frames, wins, lambda k, st, s01: got.__setitem__(k, (st.clone(), s01.clone())), batch=2, reuse_frames=False)
    assert n == 3 and sorted(got) == [0, 1, 2]
    for k, win in enumerate(wins):
        st_ref, s01_ref = module_window_u8(m, [frames[i] for i in win], N, M)
        assert torch.equal(got[k][0], st_ref.cpu()), 'St bytes of window %d' % k
        assert torch.equal(got[k][1], s01_ref.cpu()), 'S0/S1 bytes of window %d' % k
    assert got[0][0].float().std() > 5.0                               # real frames, not a constant
    del runner, m
    torch.cuda.empty_cache()


def test_config2_720p_fp16_gates_three_windows_three_t():
    """the fp16 acceptance gate on 3 windows x t in {1/8, 1/2, 7/8} instead of one window at t = 1/2.
    Reference = this repo's fp32 HIP path on the same window, which is itself pinned to the fp32 oracle at this size to
    |dPSNR| <= 1e-3 dB / PSNR > 60 dB (test_config4_720p_fp32_n5_strict_psnr) -- nine 36-second oracle forwards on the host would
    not fit the suite; the one direct fp16-vs-oracle comparison stays in test_config2_720p_fp16_n3_psnr_bounds.
    Gate: PSNR(fp16, fp32) >= 44 dB and |dPSNR vs pseudo-GT| <= 5e-3 dB for St, S0 and S1 of the last recursion."""
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 720, 1280, 3, 8
    m16, m32 = _model(torch.float16), _model(torch.float32)
    r16 = WindowRunner(m16, h, w, n_tst=N, mfi=M)
    r32 = WindowRunner(m32, h, w, n_tst=N, mfi=M, n_ctx=1)
    worst = (1e9, 0.0)
    for seed in (11, 12, 13):
        x = synthetic_window(h, w, seed).to(DEV)
        st16, s16 = [t.cpu().numpy().copy() for t in r16.run_window(x)]
        st32, s32 = [t.cpu().numpy().copy() for t in r32.run_window(x)]
        gt = x[0, :, 0].cpu().numpy()
        for j in (0, 3, 6):                                          # t = 1/8, 1/2, 7/8
            ps, dps = O.psnr(st16[j], st32[j]), O.psnr(st16[j], gt) - O.psnr(st32[j], gt)
            worst = (min(worst[0], ps), max(worst[1], abs(dps)))
            assert np.isfinite(st16[j]).all() and ps >= 44.0 and abs(dps) <= 5e-3, (seed, j, ps, dps)
        for i in range(2):                                           # deblurred frames (kept from the first time instant)
            ps, dps = O.psnr(s16[i], s32[i]), O.psnr(s16[i], gt) - O.psnr(s32[i], gt)
            assert ps >= 44.0 and abs(dps) <= 5e-3, (seed, 'S%d' % i, ps, dps)
    print('720p fp16 vs fp32 HIP path, 3 windows x 3 t: worst PSNR %.2f dB, worst |dPSNR vs pseudo-GT| %.4f dB' % worst)
    del r16, r32, m16, m32
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------
# configs[4]: 1080p -> 1088x1920, x16 (15 time instants, t_schedule(16)), N_tst = 3, fp16
# ------------------------------------------------------------------------------------------------------
def test_config5_1080p_x16_runner_properties_and_psnr():
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 1080, 1920, 3, 16
    m = _model(torch.float16)
    xs = [synthetic_window(h, w, 50 + i).to(DEV) for i in range(2)]
    runner = WindowRunner(m, h, w, n_tst=N, mfi=M)
    assert runner.engine.H == 1088 and runner.engine.W == 1920 and len(runner.ts) == 15
    assert np.allclose(runner.ts, np.arange(1, 16) / 16.0)
    st_a, s01_a = runner.run_windows(xs)
    st_a, s01_a = st_a.clone(), s01_a.clone()
    st_b, s01_b = runner.run_windows(xs)
    torch.cuda.synchronize()
    assert tuple(st_a.shape) == (2, 15, 3, h, w)
    assert torch.isfinite(st_a).all() and torch.isfinite(s01_a).all()
    assert torch.equal(st_a, st_b) and torch.equal(s01_a, s01_b)                   # run-to-run bit-identical
    # the pipelined scheduler == one pad -> forward -> crop per (window, t), on three t of each window
    ts = t_schedule(M)
    for wi in range(2):
        for k in (0, 7, 14):
            ref = pad_forward_crop(m, xs[wi], torch.tensor([[float(ts[k])]], device=DEV), N)
            assert torch.equal(st_a[wi, k], ref[1][N - 1][2][0]), (wi, k)
            if k == 0:
                assert torch.equal(s01_a[wi, 0], ref[1][N - 1][0][0]) and torch.equal(s01_a[wi, 1], ref[1][N - 1][1][0])
    # one time instant against the fp32 oracle on the padded 1088x1920 window (t = 9/16)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    xp = torch.nn.functional.pad(xs[0].cpu().reshape(1, 12, h, w), [0, 0, 0, 8], mode='reflect').reshape(1, 3, 4, 1088, w)
    tv = torch.tensor([[float(ts[8])]])
    with torch.no_grad():
        ref = O.forward(synthetic_state_dict(0), xp, tv, N)
    exp = ref[1][N - 1][2][0, :, :h].numpy()
    got = st_a[0, 8].cpu().numpy()
    gt = xs[0][0, :, 0].cpu().numpy()
    ps, dps = O.psnr(got, exp), O.psnr(got, gt) - O.psnr(exp, gt)
    print('1080p x16 fp16 t=9/16: PSNR vs fp32 oracle %.2f dB, dPSNR vs pseudo-GT %+.4f dB' % (ps, dps))
    record_fp16_margin('config5_1080p_x16_seed50_t9/16', 2, ps, dps, size='1088x1920', n_tst=3, t=9 / 16, seed=50)
    assert ps >= 44.0 and abs(dps) <= 5e-3


# ------------------------------------------------------------------------------------------------------
# integer index maps at the full widths (H = 8 rows is enough: the round trip is per axis)
# ------------------------------------------------------------------------------------------------------
def _wide_flows(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    xs = torch.arange(W).view(1, W).float()
    ys = torch.arange(H).view(H, 1).float()
    fams = {
        'zeros': torch.zeros(2, H, W),                                        # pure round-trip flips (F11)
        'ints': torch.randint(-20, 21, (2, H, W), generator=g).float(),
        'halves': torch.randint(-20, 21, (2, H, W), generator=g).float() + 0.5,
        'smooth': torch.nn.functional.avg_pool2d(torch.randn(1, 2, H + 8, W + 8, generator=g) * 25, 9, 1)[0],
        'large': torch.randn(2, H, W, generator=g) * (2.0 * W),
    }
    e = torch.zeros(2, H, W)                                                  # samples landing exactly on W-1 / H-1 / -1 / 0
    e[0] = torch.where((ys % 4) == 0, (W - 1) - xs, torch.where((ys % 4) == 1, -1 - xs, -xs))
    e[1] = torch.where((xs % 4) == 0, (H - 1) - ys, torch.where((xs % 4) == 1, -1 - ys, -ys))
    fams['edges'] = e
    return {k: v.contiguous() for k, v in fams.items()}


def _warp_maps_expected(flo, H, W):
    m = O.backward_warp_maps(flo)
    inb = sum((m['inb'][k].astype(np.int32) << k) for k in range(4)) | (m['valid'].astype(np.int32) << 4)
    return np.clip(m['ix0'], -4, W + 4).astype(np.int32), np.clip(m['iy0'], -4, H + 4).astype(np.int32), inb


@pytest.mark.parametrize('W', [1280, 1920])
def test_index_maps_bit_exact_at_full_width(W):
    H = 8
    lib = L.load()
    fams = _wide_flows(H, W, 100 + W)
    names = list(fams)
    # the round trip really is lossy at this width (otherwise the test would not test the emulation)
    z = O.backward_warp_maps(np.zeros((2, H, W), np.float32))
    moved = int((z['ix0'][0] != np.arange(W)).sum())
    assert moved > 100, moved
    t = torch.tensor([0.375], device=DEV)
    # ---- demfi_warp_blend: fat (NHWC fp16, C = 64) and thin (planar fp32, C = 3) ---------------------------------
    A16 = torch.tanh(torch.randn(H, W, 64, device=DEV)).half()
    A3 = torch.rand(3, H, W, device=DEV) * 2 - 1
    logit = torch.randn(H, W, device=DEV)
    for na, nb in zip(names, names[1:] + names[:1]):
        fa, fb = fams[na].to(DEV), fams[nb].to(DEV)
        exp = [_warp_maps_expected(fams[n].numpy(), H, W) for n in (na, nb)]
        for fat in (True, False):
            dbg = torch.zeros(2, 3, H * W, dtype=torch.int32, device=DEV)
            if fat:
                out = torch.zeros(H, W, 64, device=DEV, dtype=torch.float16)
                va = L.View(A16.data_ptr(), 64, W * 64, 1, 0, 0, 0)
                vo = L.View(out.data_ptr(), 64, W * 64, 1, 0, 0, 0)
                L.check(lib.demfi_warp_blend(C.byref(va), fa.data_ptr(), C.byref(va), fb.data_ptr(), logit.data_ptr(),
                                             t.data_ptr(), C.byref(vo), 64, H, W, None, dbg.data_ptr(), _stream()))
            else:
                out = torch.zeros(3, H, W, device=DEV)
                va = L.View(A3.data_ptr(), 1, W, H * W, 0, 1, 0)
                vo = L.View(out.data_ptr(), 1, W, H * W, 0, 1, 0)
                L.check(lib.demfi_warp_blend(C.byref(va), fa.data_ptr(), C.byref(va), fb.data_ptr(), logit.data_ptr(),
                                             t.data_ptr(), C.byref(vo), 3, H, W, None, dbg.data_ptr(), _stream()))
            torch.cuda.synchronize()
            for which in range(2):
                x0, y0, inb = exp[which]
                got = dbg[which].cpu().numpy().reshape(3, H, W)
                assert np.array_equal(got[2], inb), (na, nb, fat, which)
                anyin = (inb & 15) != 0
                assert np.array_equal(got[0][anyin], x0[anyin]) and np.array_equal(got[1][anyin], y0[anyin]), (na, nb, fat)
            if not fat:                                                         # values, thin path, exact fp32 steps
                ref = O.warp_blend(A3.cpu()[None], fams[na][None], A3.cpu()[None], fams[nb][None], logit.cpu()[None, None],
                                   t.cpu().view(1, 1, 1, 1))
                assert (out.cpu() - ref[0]).abs().max() < 3e-6, (na, nb)
    # ---- demfi_fgac_gather: absolute coordinates up to and beyond W-1 --------------------------------------------
    src = torch.tanh(torch.randn(H, W, 64, device=DEV)).half()
    g = torch.Generator().manual_seed(W)
    fl_abs = {'inrange': torch.rand(2, H, W, generator=g) * torch.tensor([W - 1.0, H - 1.0]).view(2, 1, 1),
              'ints': torch.stack([torch.arange(W).float().expand(H, W), torch.arange(H).float().view(H, 1).expand(H, W)]),
              'mixed': torch.randn(2, H, W, generator=g) * torch.tensor([W / 2.0, 6.0]).view(2, 1, 1)}
    for name, fl in fl_abs.items():
        fl = fl.contiguous()
        out = torch.zeros(H, W, 64, device=DEV, dtype=torch.float16)
        dbg = torch.zeros(3, H * W, dtype=torch.int32, device=DEV)
        vs = L.View(src.data_ptr(), 64, W * 64, 1, 0, 0, 0)
        vo = L.View(out.data_ptr(), 64, W * 64, 1, 0, 0, 0)
        L.check(lib.demfi_fgac_gather(C.byref(vs), fl.to(DEV).data_ptr(), C.byref(vo), 64, H, W, dbg.data_ptr(), _stream()))
        torch.cuda.synchronize()
        m = O.sample_maps(fl[0].numpy(), fl[1].numpy(), H, W)
        inb = sum((m['inb'][k].astype(np.int32) << k) for k in range(4))
        got = dbg.cpu().numpy().reshape(3, H, W)
        assert np.array_equal(got[2] & 15, inb), name
        anyin = inb != 0
        assert np.array_equal(got[0][anyin], m['ix0'][anyin].astype(np.int32)), name
        assert np.array_equal(got[1][anyin], m['iy0'][anyin].astype(np.int32)), name
    # ---- demfi_cfr_flow_align: splat target indices / masks ------------------------------------------------------
    for (na, nb), tv in zip((('smooth', 'ints'), ('halves', 'large'), ('edges', 'smooth')), (0.125, 0.5, 0.875)):
        f01, f10 = fams[na].to(DEV), fams[nb].to(DEV)
        tt = torch.tensor([tv], device=DEV)
        acc = torch.zeros(lib.demfi_cfr_workspace_bytes(H, W) // 8, dtype=torch.int64, device=DEV)
        out = torch.zeros(4, H, W, device=DEV)
        dbg = torch.zeros(2, 4, H * W, dtype=torch.int32, device=DEV)
        L.check(lib.demfi_cfr_flow_align(f01.data_ptr(), f10.data_ptr(), tt.data_ptr(), H, W, acc.data_ptr(), out.data_ptr(),
                                         dbg.data_ptr(), _stream()))
        torch.cuda.synchronize()
        t32 = np.float32(tv)
        for k, (fl, s) in enumerate(((fams[na].numpy(), t32), (fams[nb].numpy(), np.float32(1) - t32))):
            maps = O.splat_maps((fl * s).astype(np.float32), H, W)
            for c, mm in enumerate(maps):
                exp = np.where(mm['mask'], mm['row'] * W + mm['col'], -1).astype(np.int32).reshape(-1)
                assert np.array_equal(dbg[k, c].cpu().numpy(), exp), (na, nb, k, c)
        a, b = O.cfr_flow_align(fams[na][None], fams[nb][None], torch.tensor(tv).view(1, 1, 1, 1))
        ref = torch.cat([a[0], b[0]], 0)
        scale = max(1.0, float(ref.abs().max()))
        assert (out.cpu() - ref).abs().max() < 2e-5 * scale, (na, nb)
        assert int(acc.abs().max()) == 0                                         # workspace left clean
