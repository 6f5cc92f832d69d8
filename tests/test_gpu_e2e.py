"""End-to-end parity of the HIP forward on a real MI355X against the fixtures frozen from the upstream
reference (tests/golden, fp32: |dPSNR| <= 1e-3 dB, the north star's tolerance) and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window   # noqa: E402
from demfi_amd.harness import pad_forward_crop                                        # noqa: E402
from oracle import demfi_oracle as O                                                  # noqa: E402

DEV = 'cuda:0'
E2E = ['e2e_64x96_t0500_n3', 'e2e_64x96_t0125_n1', 'e2e_64x96_t0875_n2', 'e2e_32x64_t0375_n5']


def _model(dtype, sd=None):
    m = DeMFInet(HyperParams(), dtype=dtype)
    m.load_state_dict(sd or synthetic_state_dict(0))
    return m.to(DEV).eval()


def _close(got, ref, what, scale=1.0):
    """fp32 parity of a map: tiny on average; isolated pixels may move more because the splat / warp contain
    floor() decisions that flip under 1e-6 perturbations of the flows (the reference itself is only reproducible to
    that level across conv algorithms).  The acceptance criterion proper is the PSNR bound asserted next to it."""
    d = np.abs(got - ref)
    assert np.median(d) < 2e-5 * scale, (what, np.median(d))
    assert np.percentile(d, 90) < 2e-4 * scale, (what, np.percentile(d, 90))
    assert d.mean() < 5e-3 * scale, (what, d.mean())


@pytest.fixture(scope='module')
def model32():
    return _model(torch.float32)


@pytest.fixture(scope='module')
def model16():
    return _model(torch.float16)


def test_fp32_matches_reference_goldens(golden_dir, model32):
    for name in E2E:
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        N = int(g['N'])
        x = synthetic_window(int(g['H']), int(g['W']), int(g['seed']))
        d1, fin, flows, occs, ov = model32(x.to(DEV), torch.tensor([[float(g['t'])]], device=DEV), N)
        assert len(fin) == N and len(flows) == N + 1 and len(occs) == N + 1
        gt = x[0, :, 0].numpy()
        for i in range(3):
            assert tuple(d1[i].shape) == (1, 3, int(g['H']), int(g['W']))
            _close(d1[i][0].cpu().numpy(), g['d1'][i], name)
            for it in range(N):
                got = fin[it][i][0].cpu().numpy()
                _close(got, g['finals'][it, i], (name, it, i))
                assert abs(O.psnr(got, gt) - O.psnr(g['finals'][it, i], gt)) <= 1e-3      # north-star criterion
                assert O.psnr(got, g["finals"][it, i]) > 60.0   # 8-bit PSNR sanity bound; isolated floor() flips cost a few 1-level pixels
        for i in range(N + 1):
            _close(flows[i][0].cpu().numpy(), g['flows'][i], name, scale=10.0)
            _close(occs[i][0].cpu().numpy(), g['occs'][i], name)
        assert np.array_equal(ov[0].cpu().numpy(), g['overlay'])


def test_harness_pad_crop(golden_dir, model32):
    g = np.load(os.path.join(golden_dir, 'harness_50x70_t0625_n1.npz'))
    x = synthetic_window(50, 70, 5)
    d1, fin, flows, occs, ov = pad_forward_crop(model32, x.to(DEV), torch.tensor([[0.625]], device=DEV), 1)
    assert tuple(fin[0][2].shape) == (1, 3, 50, 70)
    for i in range(3):
        _close(fin[0][i][0].cpu().numpy(), g['finals'][0, i], 'harness')
    _close(flows[1][0].cpu().numpy(), g['flows'][1], 'harness', scale=10.0)


def test_fp32_vs_oracle_larger_frame_and_determinism(model32):
    sd = synthetic_state_dict(0)
    H, W, N = 96, 160, 3
    x = synthetic_window(H, W, 7)
    t = torch.tensor([[0.25]])
    a = model32(x.to(DEV), t.to(DEV), N)
    b = model32(x.to(DEV), t.to(DEV), N)
    with torch.no_grad():
        ref = O.forward(sd, x, t, N)
    gt = x[0, :, 1].numpy()
    for i in range(3):
        assert torch.equal(a[1][N - 1][i], b[1][N - 1][i])                      # run-to-run bit identical
        got = a[1][N - 1][i][0].cpu().numpy()
        assert abs(O.psnr(got, gt) - O.psnr(ref[1][N - 1][i][0].numpy(), gt)) <= 1e-3
    _close(a[2][N][0].cpu().numpy(), ref[2][N][0].numpy(), 'flows', scale=10.0)


def test_forward_window_equals_forward(model32):
    H, W, N = 64, 64, 2
    x = synthetic_window(H, W, 8).to(DEV)
    ts = [0.125, 0.5, 0.875]
    win = model32.forward_window(x, ts, N)
    for tv, w in zip(ts, win):
        single = model32(x, torch.tensor([[tv]], device=DEV), N)
        for i in range(3):
            assert torch.equal(w[1][N - 1][i], single[1][N - 1][i])
        assert torch.equal(w[2][N], single[2][N])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_batch_of_same_window_runs_the_batched_plan_bit_identically(dtype):
    """DeMFInet.py:51: B is a real batch dimension.  A batch whose items are one window at B time instants goes through the
    batched per-t plan (trunk once, convolutions over batch x B: VERDICT r3 missing #2); every returned tensor equals the one of
    B separate forward calls bit for bit."""
    m = DeMFInet(HyperParams(), dtype=dtype)
    m.load_state_dict(synthetic_state_dict(0))
    m = m.to(DEV).eval()
    H, W, N, B = 64, 96, 2, 3
    x1 = synthetic_window(H, W, 21).to(DEV)
    t = torch.tensor([[0.125], [0.5], [0.875]], device=DEV)
    # a stride-0 view is detected from the layout alone (no device sync, no read of the input); a materialised copy needs the
    # caller's word (same_window=True); without it the items run as distinct windows -- same results, B trunks
    for xb, kw in ((x1.expand(B, -1, -1, -1, -1), {}), (x1.repeat(B, 1, 1, 1, 1), {'same_window': True}), (x1.repeat(B, 1, 1, 1, 1), {'same_window': False})):
        m._engines.pop((H, W, dtype, 'batch'), None)
        d1, fin, flows, occs, ov = m(xb, t, N, **kw)
        batched = (H, W, dtype, 'batch') in m._engines
        assert batched == (xb.stride(0) == 0 or bool(kw.get('same_window')))
        if batched:
            assert m._engines[(H, W, dtype, 'batch')].n_ctx == B
        assert tuple(fin[N - 1][2].shape) == (B, 3, H, W) and tuple(ov.shape) == (B, 3, H, W)
        for b in range(B):
            s = m(x1, t[b:b + 1], N)
            for i in range(3):
                assert torch.equal(d1[i][b:b + 1], s[0][i])
                for it in range(N):
                    assert torch.equal(fin[it][i][b:b + 1], s[1][it][i])
            for it in range(N + 1):
                assert torch.equal(flows[it][b:b + 1], s[2][it]) and torch.equal(occs[it][b:b + 1], s[3][it])
            assert torch.equal(ov[b:b + 1], s[4])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_batch_of_distinct_windows_pipelines_trunks_bit_identically(dtype):
    """DeMFInet.py:51, VERDICT r4 missing #2: a batch of DIFFERENT windows runs its trunks on a side stream over two trunk buffer
    sets, beside the per-t segments of the previous item; every returned tensor equals the one of B separate calls bit for bit."""
    m = DeMFInet(HyperParams(), dtype=dtype)
    m.load_state_dict(synthetic_state_dict(0))
    m = m.to(DEV).eval()
    H, W, N, B = 64, 96, 2, 4
    x = torch.cat([synthetic_window(H, W, 30 + b) for b in range(B)], 0).to(DEV)
    t = torch.tensor([[0.25], [0.5], [0.75], [0.125]], device=DEV)
    d1, fin, flows, occs, ov = m(x, t, N, same_window=False)
    assert m._engines[(H, W, dtype)].n_trunk >= 2 and tuple(fin[N - 1][2].shape) == (B, 3, H, W)
    for b in range(B):
        s = m(x[b:b + 1], t[b:b + 1], N)
        for i in range(3):
            assert torch.equal(d1[i][b:b + 1], s[0][i])
            for it in range(N):
                assert torch.equal(fin[it][i][b:b + 1], s[1][it][i])
        for it in range(N + 1):
            assert torch.equal(flows[it][b:b + 1], s[2][it]) and torch.equal(occs[it][b:b + 1], s[3][it])
        assert torch.equal(ov[b:b + 1], s[4])
    assert not torch.equal(fin[N - 1][2][0], fin[N - 1][2][1])


def test_batch_of_two_and_non_shared_fgac():
    hp = HyperParams(shared_FGAC_flag=False)
    sd = synthetic_state_dict(3, hp)
    m = DeMFInet(hp, dtype=torch.float32)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = torch.cat([synthetic_window(32, 32, 6), synthetic_window(32, 32, 16)], 0)
    t = torch.tensor([[0.5], [0.25]])
    out = m(x.to(DEV), t.to(DEV), 1, same_window=False)
    assert tuple(out[1][0][2].shape) == (2, 3, 32, 32)
    with torch.no_grad():
        for b in range(2):
            ref = O.forward(sd, x[b:b + 1], t[b:b + 1], 1, shared_fgac=False)
            _close(out[1][0][2][b].cpu().numpy(), ref[1][0][2][0].numpy(), 'nonshared')


def test_fp16_psnr_against_fp32_reference(golden_dir, model16):
    """No fp16 oracle exists (the reference crashes under .half()): fp16 is stated against the fp32
    reference fixtures.  With synthetic random weights the network is chaotic around floor() decisions of the
    splat, so the bound is loose; the number itself is printed for the record."""
    for name in E2E[:2]:
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        N = int(g['N'])
        x = synthetic_window(int(g['H']), int(g['W']), int(g['seed']))
        d1, fin, flows, occs, ov = model16(x.to(DEV), torch.tensor([[float(g['t'])]], device=DEV), N)
        ps = [O.psnr(fin[N - 1][i][0].cpu().numpy(), g['finals'][N - 1, i]) for i in range(3)]
        print('fp16 vs fp32-reference PSNR (S0,S1,St) %s: %.2f %.2f %.2f dB' % (name, *ps))
        assert all(torch.isfinite(z).all() for z in fin[N - 1])
        assert min(ps) > 30.0


def test_window_runner_graph_replay_matches_module(model16):
    """The x M window scheduler (reflect pad into the engine, trunk once, hipGraph replay per t) returns exactly what
    M-1 separate pad -> forward -> crop calls of the module return."""
    from demfi_amd.harness import t_schedule
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 50, 70, 2, 4
    x = synthetic_window(h, w, 11).to(DEV)
    runner = WindowRunner(model16, h, w, n_tst=N, mfi=M, use_graph=True)
    for _ in range(2):                                     # second call replays the captured graphs
        st, s01 = runner.run_window(x)
        torch.cuda.synchronize()
        for k, tv in enumerate(t_schedule(M)):
            ref = pad_forward_crop(model16, x, torch.tensor([[float(tv)]], device=DEV), N)
            assert torch.equal(st[k], ref[1][N - 1][2][0]), k
            if k == 0:
                assert torch.equal(s01[0], ref[1][N - 1][0][0]) and torch.equal(s01[1], ref[1][N - 1][1][0])


@pytest.mark.parametrize('batched', [True, False])
def test_window_runner_pipelined_windows_match_module(model16, batched, monkeypatch):
    """run_windows (trunk of window w+1 under the time instants of window w; the 7 time instants of a window either as ONE
    launch sequence batched over 7 per-t contexts -- the default -- or one graph per time instant on five streams) returns
    bit-identical frames to one forward per (window, t)."""
    from demfi_amd.harness import t_schedule
    from demfi_amd.runner import WindowRunner
    monkeypatch.setenv('DEMFI_TB', '1' if batched else '0')
    h, w, N, M = 40, 72, 2, 8
    xs = [synthetic_window(h, w, 30 + i).to(DEV) for i in range(5)]
    runner = WindowRunner(model16, h, w, n_tst=N, mfi=M, use_graph=True)
    assert runner.n_trunk == (3 if batched else 2) and runner.tb == batched and runner.n_ctx == (7 if batched else 5)
    for rep in range(2):
        st, s01 = runner.run_windows(xs)
        torch.cuda.synchronize()
        for wi, x in enumerate(xs):
            for k, tv in enumerate(t_schedule(M)):
                if rep == 1 and (wi + k) % 3:              # second pass: spot check
                    continue
                ref = pad_forward_crop(model16, x, torch.tensor([[float(tv)]], device=DEV), N)
                assert torch.equal(st[wi, k], ref[1][N - 1][2][0]), (wi, k)
                if k == 0:
                    assert torch.equal(s01[wi, 0], ref[1][N - 1][0][0]) and torch.equal(s01[wi, 1], ref[1][N - 1][1][0])
    # the single-window entry point shares the contexts with the pipelined one
    st1, _ = runner.run_window(xs[3])
    torch.cuda.synchronize()
    assert torch.equal(st1, st[3])


def test_full_size_720p_fp16_properties(model16):
    """BASELINE config 2 shape (720p -> 736x1280, N_tst=3): size-independent properties -- finite outputs, run-to-run
    bit-identical results (deterministic splat + fixed MFMA accumulation order, also across the persistent kernel's
    dynamic tile walk), occlusion maps in [0,1], and D2 frames = D1 frames + residual (same S0p/S1p across t)."""
    x = synthetic_window(720, 1280, 21).to(DEV)
    a = pad_forward_crop(model16, x, torch.tensor([[0.375]], device=DEV), 3)
    b = pad_forward_crop(model16, x, torch.tensor([[0.375]], device=DEV), 3)
    c = pad_forward_crop(model16, x, torch.tensor([[0.625]], device=DEV), 3)
    for i in range(3):
        assert tuple(a[1][2][i].shape) == (1, 3, 720, 1280)
        assert torch.isfinite(a[1][2][i]).all()
        assert torch.equal(a[1][2][i], b[1][2][i])
    for o in a[3]:
        assert float(o.min()) >= 0.0 and float(o.max()) <= 1.0
    assert torch.equal(a[4], c[4])                                  # overlay does not depend on t
    assert not torch.equal(a[1][2][2], c[1][2][2])                  # St does


def test_uint8_io_bit_exact_and_runner_u8(model16):
    """uint8 BGR frames in (normalise + reflect pad fused), uint8 frames out (crop +
    denorm255 + truncation fused) -- both bit-identical to the reference's host arithmetic."""
    import ctypes as C
    from demfi_amd import _lib as L
    from demfi_amd.harness import t_schedule
    from demfi_amd.runner import WindowRunner
    lib = L.load()
    h, w, H, W = 50, 70, 64, 96
    g = torch.Generator().manual_seed(5)
    frames = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for _ in range(4)]
    dev = [f.to(DEV) for f in frames]
    x = torch.zeros(3, 4, H, W, device=DEV)
    ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in dev])
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.demfi_u8_to_window(ptrs, h, w, x.data_ptr(), H, W, st))
    ref = O.frames_u8_to_tensor([f.numpy() for f in frames])                       # [3,4,h,w]
    refp = torch.nn.functional.pad(ref.reshape(1, 12, h, w), [0, W - w, 0, H - h], mode='reflect').reshape(3, 4, H, W)
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), refp)
    fr = (torch.randn(3, H, W, generator=g) * 0.8).to(DEV)
    out = torch.zeros(h, w, 3, dtype=torch.uint8, device=DEV)
    L.check(lib.demfi_frame_to_u8(fr.data_ptr(), out.data_ptr(), h, w, H, W, st))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.frame_to_u8(fr.cpu().numpy()[:, :h, :w]))
    # fused ingest: x + space-to-depth + overlay in one pass == the three separate kernels
    for dtype, dt in ((torch.float16, L.F16), (torch.float32, L.F32)):
        x2 = torch.zeros(3, 4, H, W, device=DEV)
        s2d2 = torch.zeros(H // 2, W // 2, 48, device=DEV, dtype=dtype)
        ov2 = torch.zeros(3, H, W, device=DEV)
        L.check(lib.demfi_u8_ingest(ptrs, h, w, x2.data_ptr(), s2d2.data_ptr(), ov2.data_ptr(), dt, H, W, st))
        s2d1 = torch.zeros_like(s2d2)
        ov1 = torch.zeros_like(ov2)
        L.check(lib.demfi_space_to_depth(x.data_ptr(), s2d1.data_ptr(), dt, H, W, st))
        L.check(lib.demfi_overlay_mean(x.data_ptr(), ov1.data_ptr(), H, W, st))
        torch.cuda.synchronize()
        assert torch.equal(x2, x) and torch.equal(s2d2, s2d1) and torch.equal(ov2, ov1)
    # the uint8 runner == float runner on the same window, then quantised
    N, M = 2, 4
    runner = WindowRunner(model16, h, w, n_tst=N, mfi=M)
    stf, s01f = runner.run_window(ref[None].to(DEV))
    stf, s01f = stf.clone(), s01f.clone()
    stu, s01u = runner.run_window_u8(dev)
    torch.cuda.synchronize()
    for k in range(M - 1):
        assert np.array_equal(stu[k].cpu().numpy(), O.frame_to_u8(stf[k].cpu().numpy()))
    assert np.array_equal(s01u[1].cpu().numpy(), O.frame_to_u8(s01f[1].cpu().numpy()))
    assert np.array_equal(s01u[0].cpu().numpy(), O.frame_to_u8(s01f[0].cpu().numpy()))
    # the uint8 sink of the last layer must not leak into a later float run on the same engine (and vice versa)
    stf2, _ = runner.run_window(ref[None].to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(stf2, stf)
    out_m = model16(torch.nn.functional.pad(ref.reshape(1, 12, h, w), [0, W - w, 0, H - h], mode='reflect').reshape(1, 3, 4, H, W).to(DEV),
                    torch.tensor([[float(t_schedule(M)[1])]], device=DEV), N)
    assert torch.equal(out_m[1][N - 1][2][0, :, :h, :w], stf[1])
    # fp32 engine: no sink epilogue, the separate egress kernel is used -- same bytes as the oracle
    m32 = _model(torch.float32)
    r32 = WindowRunner(m32, h, w, n_tst=1, mfi=2)
    f32, _ = r32.run_window(ref[None].to(DEV))
    f32 = f32.clone()
    u32, _ = r32.run_window_u8(dev)
    torch.cuda.synchronize()
    assert np.array_equal(u32[0].cpu().numpy(), O.frame_to_u8(f32[0].cpu().numpy()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_ragged_frame_size_multiple_of_8_only(dtype):
    """The model itself only needs H, W % 8 == 0 (UNet, DeMFInet.py:575-603); tiles of 8x32 are then partial in x
    (W = 72 -> 2.25 tiles) and the persistent kernel walks fewer tiles than CUs."""
    sd = synthetic_state_dict(0)
    m = _model(dtype, sd)
    x = synthetic_window(40, 72, 31)
    t = torch.tensor([[0.375]])
    out = m(x.to(DEV), t.to(DEV), 2)
    with torch.no_grad():
        ref = O.forward(sd, x, t, 2)
    for i in range(3):
        got = out[1][1][i][0].cpu().numpy()
        assert np.isfinite(got).all()
        if dtype == torch.float32:
            _close(got, ref[1][1][i][0].numpy(), 'ragged')
        else:
            assert O.psnr(got, ref[1][1][i][0].numpy()) > 30.0


def test_final_only_runner_delivers_the_same_frames(model16):
    """WindowRunner(final_only=True) skips the warp + D2 tail of the recursions before the last one (their frames are outputs
    only, DeMFInet.py:146-165): the delivered frames -- fp32 and through the uint8 sink -- are bit-identical, and the earlier
    recursions' frames are indeed not produced."""
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 40, 72, 3, 8
    x = synthetic_window(h, w, 91).to(DEV)
    full = WindowRunner(model16, h, w, n_tst=N, mfi=M)
    st_a, s01_a = full.run_window(x)
    st_a, s01_a = st_a.clone(), s01_a.clone()
    g = torch.Generator().manual_seed(3)
    frames = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(4)]
    u_a, us_a = [t.clone() for t in full.run_window_u8(frames)]
    torch.cuda.synchronize()
    del full
    fo = WindowRunner(model16, h, w, n_tst=N, mfi=M, final_only=True)
    assert fo.tb and fo.final_only
    eng = fo.engine
    for row in eng._ctxs:
        for ctx in row:
            ctx['finals'].zero_()
    st_b, s01_b = fo.run_window(x)
    torch.cuda.synchronize()
    assert torch.equal(st_a, st_b) and torch.equal(s01_a, s01_b)
    assert float(eng._ctxs[0][0]['finals'][0].abs().max()) == 0.0 and float(eng._ctxs[0][0]['finals'][N - 1].abs().max()) > 0.0
    u_b, us_b = fo.run_window_u8(frames)
    torch.cuda.synchronize()
    assert torch.equal(u_a, u_b) and torch.equal(us_a, us_b)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_visualisation_and_training_return_tuples(golden_dir, dtype):
    """Round 6, SURVEY 8(f4): args.visualization_flag = True -> the 7-tuple of DeMFInet.py:174-176 (blending_weights with FGAC's six
    maps per direction + [flow_01, flow_10], difference_maps), is_training = True -> the 7-tuple of 170-172 (difference_maps,
    flow_t0_t1_predictions); fixture from the unpatched reference.  The five leading members are those of the plain forward."""
    g = np.load(os.path.join(golden_dir, 'extras_64x96_t0500_n1.npz'))
    H, W = int(g['H']), int(g['W'])
    x = synthetic_window(H, W, int(g['seed'])).to(DEV)
    t = torch.tensor([[float(g['t'])]], device=DEV)
    mv = DeMFInet(HyperParams(visualization_flag=True), dtype=dtype)
    mv.load_state_dict(synthetic_state_dict(0))
    mv = mv.to(DEV).eval()
    out = mv(x, t, 1)
    assert len(out) == 7
    bw, diffs = out[5], out[6]
    assert len(bw) == 5 and len(diffs) == 4 and len(bw[0]) == 6 and len(bw[4]) == 2
    # fp32: conv summation order (1e-5 on the gates), amplified by 1 / (max - min) in the normalised maps; fp16: fp16 features
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for b in range(2):
        for k in range(6):
            assert tuple(bw[b][k].shape) == (1, 1, H, W)
            d = np.abs(bw[b][k][0, 0].cpu().numpy() - g['bw'][b, k])
            assert d.max() < tol and np.median(d) < tol / 10, (b, k, d.max(), np.median(d))
            assert torch.equal(bw[b][k], bw[b + 2][k])
        assert np.abs(diffs[b][0, 0].cpu().numpy() - g['diff'][b]).max() < tol
        assert torch.equal(diffs[b], diffs[b + 2])
        if dtype == torch.float32:
            assert float(bw[b][4].min()) == 0.0 and float(bw[b][4].max()) == 1.0
    assert np.median(np.abs(bw[4][0][0].cpu().numpy() - g['flow_01'])) < (2e-4 if dtype == torch.float32 else 5e-2)
    plain = _model(dtype)
    ref5 = plain(x, t, 1)
    assert torch.equal(out[1][0][2], ref5[1][0][2]) and torch.equal(out[0][1], ref5[0][1])
    # training tuple from a model built WITHOUT the flag: an engine with the extras is built on demand
    tr = plain(x, t, 1, True)
    assert len(tr) == 7 and len(tr[5]) == 4 and len(tr[6]) == 1 and len(tr[6][0]) == 2
    assert np.abs(tr[5][1][0, 0].cpu().numpy() - g['train_diff'][1]).max() < tol
    assert torch.equal(tr[1][0][2], ref5[1][0][2])
    assert torch.equal(tr[6][0][0], ref5[2][0][:, 0:2]) and torch.equal(tr[6][0][1], ref5[2][0][:, 2:4])
    if dtype == torch.float32:
        assert np.median(np.abs(tr[6][0][0][0].cpu().numpy() - g['train_rflow'][0])) < 2e-4
    # a batch of the same window: the maps are replicated per item
    ob = mv(x.expand(2, -1, -1, -1, -1), torch.tensor([[0.5], [0.25]], device=DEV), 1)
    assert tuple(ob[5][0][2].shape) == (2, 1, H, W) and torch.equal(ob[5][0][2][0], bw[0][2][0]) and torch.equal(ob[6][1][1], diffs[1][0])


def test_second_weight_regime_fp32_goldens_and_fp16_margin(golden_dir):
    """Round 6: small flows (<= 3 px) and unsaturated occlusion maps -- synthetic_state_dict(flow_gain=0.3)
    -- against fixtures from the reference: fp32 |dPSNR| <= 1e-3 dB; fp16 stated against the fp32 fixtures, margins recorded."""
    from tests.conftest import record_fp16_margin
    sd = synthetic_state_dict(0, flow_gain=0.3)
    m32, m16 = _model(torch.float32, sd), _model(torch.float16, sd)
    for name in ('e2e_smallflow_64x96_t0500_n3', 'e2e_smallflow_64x96_t0125_n2'):
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        N = int(g['N'])
        x = synthetic_window(int(g['H']), int(g['W']), int(g['seed']))
        t = torch.tensor([[float(g['t'])]], device=DEV)
        d1, fin, flows, occs, ov = m32(x.to(DEV), t, N)
        gt = x[0, :, 0].numpy()
        for it in range(N):
            for i in range(3):
                got = fin[it][i][0].cpu().numpy()
                _close(got, g['finals'][it, i], (name, it, i))
                assert abs(O.psnr(got, gt) - O.psnr(g['finals'][it, i], gt)) <= 1e-3
                assert O.psnr(got, g['finals'][it, i]) > 60.0
        for i in range(N + 1):
            _close(flows[i][0].cpu().numpy(), g['flows'][i], name, scale=10.0)
            _close(occs[i][0].cpu().numpy(), g['occs'][i], name)
        f16 = m16(x.to(DEV), t, N)[1]
        for i in range(3):
            got = f16[N - 1][i][0].cpu().numpy()
            ps, dps = O.psnr(got, g['finals'][N - 1, i]), O.psnr(got, gt) - O.psnr(g['finals'][N - 1, i], gt)
            print('%s fp16 frame %d: PSNR vs the reference fixture %.2f dB, dPSNR vs pseudo-GT %+.4f dB' % (name, i, ps, dps))
            record_fp16_margin('regime2_' + name, i, ps, dps, size='64x96', n_tst=N, t=float(g['t']), seed=int(g['seed']), flow_gain=0.3)
            assert np.isfinite(got).all() and ps >= 40.0


def test_materialised_same_window_batch_warns_once():
    """a batch of materialised copies without the caller's word runs as distinct windows (identical results) and says so once."""
    import warnings
    m = _model(torch.float16)
    x = synthetic_window(32, 64, 3).to(DEV).repeat(2, 1, 1, 1, 1)
    t = torch.tensor([[0.25], [0.75]], device=DEV)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        a = m(x, t, 1)
        b = m(x, t, 1)
    assert sum('DIFFERENT windows' in str(i.message) for i in w) == 1
    c = m(x, t, 1, same_window=True)
    assert torch.equal(a[1][0][2], c[1][0][2]) and torch.equal(a[1][0][2], b[1][0][2])
