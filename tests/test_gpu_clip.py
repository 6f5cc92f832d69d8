"""Clip edge on a real MI355X: host-to-host uint8 pipeline, folder run through the PNG codec, sharding, on-GPU PSNR/SSIM."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window   # noqa: E402
from demfi_amd import clipio                                                          # noqa: E402
from demfi_amd.clip import ClipRunner, output_names, window_list                      # noqa: E402
from demfi_amd.metrics import FrameEvaluator, u8_frame_to_tensor                      # noqa: E402
from oracle import demfi_oracle as O                                                  # noqa: E402

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def model16():
    m = DeMFInet(HyperParams(), dtype=torch.float16)
    m.load_state_dict(synthetic_state_dict(0))
    return m.to(DEV).eval()


def _clip(h, w, n, seed=0):
    """n uint8 BGR frames of a moving pattern (host)."""
    base = synthetic_window(h + 2 * n, w + 2 * n, seed)[0, :, 0]                        # [3,H,W] in [-1,1]
    out = []
    for i in range(n):
        f = base[:, i:i + h, 2 * i:2 * i + w]
        out.append(np.ascontiguousarray(((f.permute(1, 2, 0).numpy() + 1) * 127.5).clip(0, 255).astype(np.uint8)))
    return out


def test_run_clip_u8_equals_per_window_runs(model16):
    from demfi_amd.runner import WindowRunner
    h, w, N, M = 50, 70, 2, 4
    frames = _clip(h, w, 9, 1)
    wins = window_list(len(frames))
    runner = WindowRunner(model16, h, w, n_tst=N, mfi=M)
    ref = []
    dev = [torch.from_numpy(f).to(DEV) for f in frames]
    for wdw in wins:
        st, s01 = runner.run_window_u8([dev[i] for i in wdw])
        torch.cuda.synchronize()
        ref.append((st.cpu().clone(), s01.cpu().clone()))
    host = [torch.from_numpy(f).pin_memory() for f in frames]
    for reuse in (True, False):
        got = {}
        n = runner.run_clip_u8(host, wins, lambda k, st, s01: got.__setitem__(k, (st.clone(), s01.clone())), batch=2, reuse_frames=reuse)
        assert n == len(wins) and sorted(got) == list(range(len(wins)))
        for k in range(len(wins)):
            assert torch.equal(got[k][0], ref[k][0]) and torch.equal(got[k][1], ref[k][1]), (reuse, k)


def test_folder_run_png_in_png_out_and_sharding(model16, tmp_path):
    h, w, N, M = 40, 72, 1, 4
    frames = _clip(h, w, 7, 2)
    scene = tmp_path / 'scene1'
    scene.mkdir()
    names = []
    for i, f in enumerate(frames):
        names.append(str(scene / ('%05d.png' % i)))
        clipio.write_frame(names[-1], f)
    # single rank
    cr = ClipRunner(model16, h, w, n_tst=N, mfi=M, batch=2)
    out1 = str(tmp_path / 'out1')
    nwin, nfr = cr.run_folder(str(scene), out1)
    assert nwin == 4 and nfr == 4 * (M - 1) + 4 + 1            # St of every window, S0 of every window, S1 of the last one only
    assert cr.last_decode_peak <= 2 * 2 + 3 + 4 + 2            # streamed decode: O(ahead) frames alive, not the clip
    exp_files = set()
    for st_names, s0, s1 in output_names(names, M):
        exp_files.update(st_names + [s0, s1])
    assert set(os.listdir(out1)) == exp_files
    # frames written == frames the runner returns for the same windows
    got = {}
    cr.run_frames(frames, lambda k, st, s01: got.__setitem__(k, (st.numpy().copy(), s01.numpy().copy())))
    onames = output_names(names, M)
    for k, (st_names, s0, s1) in enumerate(onames):
        for j, nm in enumerate(st_names):
            assert np.array_equal(clipio.read_frame(os.path.join(out1, nm)), got[k][0][j])
        # deblurred frames: the file named after B0 holds THIS window's S0 (what the reference's sequential loop leaves), the
        # clip's last B1 file the last window's S1 -- deterministic, one writer per file
        assert np.array_equal(clipio.read_frame(os.path.join(out1, s0)), got[k][1][0])
    assert np.array_equal(clipio.read_frame(os.path.join(out1, onames[-1][2])), got[len(onames) - 1][1][1])
    # two ranks (run one after the other on this GPU): disjoint windows, same files in the end
    out2 = str(tmp_path / 'out2')
    tot = 0
    for rank in range(2):
        crr = ClipRunner(model16, h, w, n_tst=N, mfi=M, batch=2, world=2, rank=rank)
        nw, _ = crr.run_folder(str(scene), out2)
        tot += nw
    assert tot == 4 and set(os.listdir(out2)) == exp_files
    for nm in exp_files:                                                # every file (St AND deblurred) has exactly one writer
        assert np.array_equal(clipio.read_frame(os.path.join(out2, nm)), clipio.read_frame(os.path.join(out1, nm)))


def test_gpu_psnr_ssim_match_reference_fixture_and_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, 'metrics_96x128.npz'))
    a, b = torch.from_numpy(g['a']).to(DEV), torch.from_numpy(g['b']).to(DEV)
    ev = FrameEvaluator(96, 128, DEV)
    # fixture: both images rounded (psnr / ssim of the reference code on np.around(denorm255_np(.)) images)
    p, s = ev(b, a, round_gt=True)                                  # pred = b, target = a (symmetric for both metrics up to order)
    assert abs(p - float(g['psnr'])) < 1e-9 and abs(s - float(g['ssim'])) < 1e-9
    p2, s2 = ev(a, a, round_gt=True)
    assert p2 == float('inf') and abs(s2 - 1.0) < 1e-12
    # test()'s convention: prediction rounded, target not (main.py:762-770) -- against the oracle
    po, so = O.eval_frame(g['b'], g['a'])
    p3, s3 = ev(b, a)
    assert abs(p3 - po) < 1e-9 and abs(s3 - so) < 1e-9
    # a crop of a larger (padded) buffer is a view: strides are honoured
    big = torch.zeros(3, 128, 160, device=DEV)
    big[:, :96, :128] = b
    p4, s4 = ev(big, a)
    assert p4 == p3 and s4 == s3
    # run-to-run bit-identical
    assert ev(b, a) == (p3, s3)


def test_gpu_eval_on_uint8_targets_full_size():
    """720p: target from an 8-bit frame through the loader's arithmetic, prediction = target + noise; vs the oracle."""
    h, w = 720, 1280
    gen = torch.Generator().manual_seed(3)
    gt_u8 = torch.randint(0, 256, (h, w, 3), generator=gen, dtype=torch.uint8)
    gt = u8_frame_to_tensor(gt_u8.to(DEV))
    assert torch.equal(gt.cpu(), O.frames_u8_to_tensor([gt_u8.numpy()])[:, 0])
    pred = (gt + 0.03 * torch.randn(gt.shape, generator=gen).to(DEV)).contiguous()
    p, s = FrameEvaluator(h, w, DEV)(pred, gt)
    po, so = O.eval_frame(pred.cpu().numpy(), gt.cpu().numpy())
    assert abs(p - po) < 1e-9 and abs(s - so) < 1e-9


def test_evaluate_scene_against_ground_truth_matches_oracle_metrics(model16, tmp_path):
    """ClipRunner.evaluate = the metric half of test() (main.py:756-838): blur folder + sharp folder in the layout of
    make_2D_dataset_Test, D1 and D2 frames of every (window, t) against the GT, averaged per time index.  Checked against the
    oracle's eval_frame on the frames the runner delivers, on a 3-window clip; two ranks reduce to the same table."""
    from demfi_amd.clip import EvalTable, deblur_time_indices, gt_names
    from demfi_amd.runner import WindowRunner
    h, w, N, M, step = 40, 72, 2, 4, 8
    blur, sharp = tmp_path / 'test_blur' / 's0', tmp_path / 'test' / 's0'
    blur.mkdir(parents=True)
    sharp.mkdir(parents=True)
    nums = [1 + step * i for i in range(6)]                              # 6 blurry frames -> 3 windows
    bframes = _clip(h, w, 6, 5)
    names = []
    for n, f in zip(nums, bframes):
        names.append(str(blur / ('%05d.png' % n)))
        clipio.write_frame(names[-1], f)
    sframes = _clip(h, w, nums[-1] + 1, 6)
    for n in range(nums[-1] + 1):
        clipio.write_frame(str(sharp / ('%05d.png' % n)), sframes[n])
    cr = ClipRunner(model16, h, w, n_tst=N, mfi=M, batch=2)
    tabs, nwin = cr.evaluate(str(blur), str(sharp), t_step_size=step)
    assert nwin == 3
    # reference: the same windows through the runner, metrics by the oracle (numpy fp64 restatement of utils.py:652-705)
    runner = WindowRunner(model16, h, w, n_tst=N, mfi=M)
    exp = {'D1': EvalTable(M), 'D2': EvalTable(M)}
    gts = gt_names(names, M, step)
    to_t = lambda f: u8_frame_to_tensor(torch.from_numpy(f).to(DEV))
    for k, win in enumerate(window_list(6)):
        x = torch.stack([to_t(bframes[i]) for i in win], 1).unsqueeze(0)
        j_s0, j_s1 = deblur_time_indices(M)
        # S0 / S1 of the time instants test() scores them at: one forward per t through the MODULE (Stage I's S0' / S1' depend on t)
        st, s01, st1, s011 = [t.cpu().numpy().copy() for t in runner.run_window(x, with_d1=True)]
        from demfi_amd.harness import reflect_pad_to_multiple
        per_t = model16.forward_window(reflect_pad_to_multiple(x, 32), runner.ts, N)
        s01 = [per_t[j_s0][1][N - 1][0][0, :, :h, :w].cpu().numpy(), per_t[j_s1][1][N - 1][1][0, :, :h, :w].cpu().numpy()]
        s011 = [per_t[j_s0][0][0][0, :, :h, :w].cpu().numpy(), per_t[j_s1][0][1][0, :, :h, :w].cpu().numpy()]
        for j in range(M - 1):
            g = to_t(clipio.read_frame(str(sharp / gts[k][0][j]))).cpu().numpy()
            exp['D1'].update('s0', j, *O.eval_frame(st1[j], g))
            exp['D2'].update('s0', j, *O.eval_frame(st[j], g))
        for i in range(2 if k == 2 else 1):                              # S1: the scene's last window only (main.py:633-645, 1053-1058)
            g = to_t(clipio.read_frame(str(sharp / gts[k][1 + i]))).cpu().numpy()
            exp['D1'].update_deblur('s0', *O.eval_frame(s011[i], g))
            exp['D2'].update_deblur('s0', *O.eval_frame(s01[i], g))
    for key in ('D1', 'D2'):
        a, b = tabs[key].summary(), exp[key].summary()
        assert a['samples'] == b['samples'] == 3 * (M - 1)
        assert np.allclose(a['per_index'], b['per_index'], rtol=0, atol=1e-8) and np.allclose(a['total'], b['total'], rtol=0, atol=1e-8)
        assert a['deblur_samples'] == b['deblur_samples'] == 3 + 1
        assert np.allclose([a['deblur'], a['deblur_total']], [b['deblur'], b['deblur_total']], rtol=0, atol=1e-8)
    assert tabs['D2'].summary()['total'][0] != tabs['D1'].summary()['total'][0]      # the two stages are different frames
    # two ranks (one after the other on this GPU) + the reduction vector = the single-rank table
    merged = {'D1': None, 'D2': None}
    for rank in range(2):
        t2, _ = ClipRunner(model16, h, w, n_tst=N, mfi=M, batch=2, world=2, rank=rank).evaluate(str(blur), str(sharp), t_step_size=step)
        for key in merged:
            v = t2[key].merge_vector(['s0'])
            merged[key] = v if merged[key] is None else [x + y for x, y in zip(merged[key], v)]
    for key in merged:
        t = EvalTable(M)
        t.merge_from(['s0'], merged[key])
        assert np.allclose(t.summary()['per_index'], tabs[key].summary()['per_index'], rtol=0, atol=1e-9)


def test_clip_cli_folder_in_folder_out_with_checkpoint(model16, tmp_path):
    """``python -m demfi_amd.clip <custom_path> --checkpoint ...`` = main.py --phase test_custom (main.py:1108-1196): scene
    folders in, ``<scene>_sharply_interpolated_xM`` folders out under the reference's names; the checkpoint file has the layout
    SaveManager writes (``state_dict_Model``).  The files equal what ClipRunner.run_folder writes with the same weights."""
    import json
    import subprocess
    import sys
    from demfi_amd import synthetic_state_dict
    h, w, M = 40, 72, 4
    root = tmp_path / 'custom'
    for s, seed in (('sceneA', 11), ('sceneB', 12)):
        (root / s).mkdir(parents=True)
        for i, f in enumerate(_clip(h, w, 5, seed)):
            clipio.write_frame(str(root / s / ('%05d.png' % i)), f)
    ck = str(tmp_path / 'DeMFInet_latest.pt')
    torch.save({'last_epoch': 0, 'state_dict_Model': synthetic_state_dict(0)}, ck)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-m', 'demfi_amd.clip', str(root), '--checkpoint', ck, '--mfi', str(M), '--n-tst', '2'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info['scenes'] == 2 and info['windows'] == 4 and info['weights'] == 'DeMFInet_latest.pt'
    assert info['png_written'] == 2 * (2 * (M - 1) + 2 + 1)         # per scene: 2 windows x (M-1) St + S0 of each window + the last S1
    cr = ClipRunner(model16, h, w, n_tst=2, mfi=M, batch=2)
    cr.run_folder(str(root / 'sceneB'), str(tmp_path / 'exp'))
    out = root / ('sceneB_sharply_interpolated_x%d' % M)
    names = sorted(os.listdir(str(tmp_path / 'exp')))
    assert sorted(os.listdir(str(out))) == names and '00001_000.png' in names and '00002_002.png' in names
    for nm in names:
        assert np.array_equal(clipio.read_frame(str(out / nm)), clipio.read_frame(str(tmp_path / 'exp' / nm))), nm
