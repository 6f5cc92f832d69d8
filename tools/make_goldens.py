"""Generate tests/golden/*.npz by running the UPSTREAM reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py

Imports /root/reference/DeMFInet.py (read-only; it never travels to the GPU box), loads this repo's
synthetic weights (demfi_amd.weights) into it and freezes inputs + outputs as small fixtures.  Fixtures
are DATA (inputs, expected outputs); no reference source is copied.  The script also prints how far the
oracle restatement (oracle/demfi_oracle.py) is from the reference on each case.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

import DeMFInet as R                                      # noqa: E402  (the upstream reference)
from demfi_amd.weights import synthetic_state_dict, synthetic_window   # noqa: E402
from oracle import demfi_oracle as O                      # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
CPU = torch.device('cpu')
ARGS = types.SimpleNamespace(gpu=0, nf=64, scale_factor=2, num_ResB_FACFB=5, num_ResB_Dec=5,
                             shared_FGAC_flag=True, visualization_flag=False)


def flow_families(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    fams = {}
    fams['zeros'] = torch.zeros(1, 2, H, W)
    fams['ints'] = torch.randint(-6, 7, (1, 2, H, W), generator=g).float()
    fams['halves'] = torch.randint(-6, 7, (1, 2, H, W), generator=g).float() + 0.5
    fams['smooth'] = torch.nn.functional.avg_pool2d(torch.randn(1, 2, H + 8, W + 8, generator=g) * 12, 9, 1)
    fams['large'] = torch.randn(1, 2, H, W, generator=g) * (2.0 * W)
    e = torch.zeros(1, 2, H, W)                              # samples landing exactly on W-1 / H-1 / -1 / 0
    xs = torch.arange(W).view(1, W).float()
    ys = torch.arange(H).view(H, 1).float()
    e[0, 0] = torch.where((ys % 4) == 0, (W - 1) - xs, torch.where((ys % 4) == 1, -1 - xs, -xs))
    e[0, 1] = torch.where((xs % 4) == 0, (H - 1) - ys, torch.where((xs % 4) == 1, -1 - ys, -ys))
    fams['edges'] = e
    fams['collide'] = torch.stack([(W // 2 - xs).expand(H, W) + 0.25, (H // 2 - ys).expand(H, W) - 0.25])[None]
    return fams


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    sd = synthetic_state_dict(0)
    net = R.DeMFInet(ARGS).eval()
    net.load_state_dict(sd)

    # ---------------- end-to-end forward (padded sizes) -------------------------------------------------
    worst = 0.0
    for tag, (H, W, seed, tval, N) in {
        'e2e_64x96_t0500_n3': (64, 96, 1, 0.5, 3),
        'e2e_64x96_t0125_n1': (64, 96, 2, 0.125, 1),
        'e2e_64x96_t0875_n2': (64, 96, 3, 0.875, 2),
        'e2e_32x64_t0375_n5': (32, 64, 4, 0.375, 5),
    }.items():
        x = synthetic_window(H, W, seed)
        t = torch.tensor([[tval]], dtype=torch.float32)
        with torch.no_grad():
            d1, fin, flows, occs, ov = net(x, t, N)
            mine = O.forward(sd, x, t, N)
        rec = dict(H=H, W=W, seed=seed, t=np.float32(tval), N=N, weight_seed=0,
                   d1=torch.stack([z[0] for z in d1]).numpy(),
                   finals=torch.stack([torch.stack([z[0] for z in f]) for f in fin]).numpy(),
                   flows=torch.stack([z[0] for z in flows]).numpy(),
                   occs=torch.stack([z[0] for z in occs]).numpy(),
                   overlay=ov[0].numpy())
        np.savez_compressed(os.path.join(OUT, tag + '.npz'), **rec)
        d = max(float((a - b).abs().max()) for a, b in zip(fin[-1], mine[1][-1]))
        d = max(d, max(float((a - b).abs().max()) for a, b in zip(flows, mine[2])))
        worst = max(worst, d)
        print('%-22s oracle-vs-reference max|diff| = %.3e' % (tag, d))

    # ---------------- harness: unpadded 50x70 -> reflect pad 64x96 -> crop -------------------------------
    x = synthetic_window(50, 70, 5)
    t = torch.tensor([[0.625]], dtype=torch.float32)
    xp = torch.nn.functional.pad(x.reshape(1, 12, 50, 70), [0, 26, 0, 14], mode='reflect').reshape(1, 3, 4, 64, 96)
    with torch.no_grad():
        d1, fin, flows, occs, ov = net(xp, t, 1)
        mine = O.pad_forward_crop(sd, x, t, 1)
    rec = dict(H=50, W=70, seed=5, t=np.float32(0.625), N=1, weight_seed=0,
               d1=torch.stack([z[0, :, :50, :70] for z in d1]).numpy(),
               finals=torch.stack([torch.stack([z[0, :, :50, :70] for z in f]) for f in fin]).numpy(),
               flows=torch.stack([z[0, :, :50, :70] for z in flows]).numpy(),
               occs=torch.stack([z[0, :, :50, :70] for z in occs]).numpy(),
               overlay=ov[0, :, :50, :70].numpy())
    np.savez_compressed(os.path.join(OUT, 'harness_50x70_t0625_n1.npz'), **rec)
    print('harness_50x70          oracle-vs-reference max|diff| = %.3e' %
          max(float((a[0, :, :50, :70] - b[0]).abs().max()) for a, b in zip(fin[-1], mine[1][-1])))

    # ---------------- kernel-level: bwarp / fwarp / CFR ----------------------------------------------------
    H, W = 24, 40
    g = torch.Generator().manual_seed(11)
    img3 = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    img8 = torch.rand(1, 8, H, W, generator=g) * 2 - 1
    rec = dict(img3=img3[0].numpy(), img8=img8[0].numpy())
    for name, flo in flow_families(H, W, 12).items():
        with torch.no_grad():
            o3 = R.bwarp(CPU, img3, flo)
            o8 = R.bwarp(CPU, img8, flo)
            iw, io = R.fwarp(CPU, flo, 0.375 * flo)
            mine3 = O.backward_warp_explicit(img3, flo)
            mw, mo = O.forward_splat(flo, 0.375 * flo)
        rec['flo_' + name] = flo[0].numpy()
        rec['bwarp3_' + name] = o3[0].numpy()
        rec['bwarp8_' + name] = o8[0].numpy()
        rec['fwarp_img_' + name] = iw[0].numpy()
        rec['fwarp_one_' + name] = io[0].numpy()
        print('bwarp/%-8s explicit-vs-reference %.3e   fwarp bit-exact: %s' %
              (name, float((o3 - mine3).abs().max()), bool(torch.equal(iw, mw) and torch.equal(io, mo))))
    fams = flow_families(H, W, 13)
    for i, (a, b, tval) in enumerate((('smooth', 'ints', 0.125), ('halves', 'smooth', 0.5),
                                      ('collide', 'edges', 0.875), ('large', 'smooth', 0.375))):
        t4 = torch.tensor(tval, dtype=torch.float32).view(1, 1, 1, 1)
        with torch.no_grad():
            r0, r1 = R.CFR_flow_t_align(CPU, fams[a].clone(), fams[b].clone(), t4)
            m0, m1 = O.cfr_flow_align(fams[a], fams[b], t4)
        rec['cfr%d_f01' % i] = fams[a][0].numpy()
        rec['cfr%d_f10' % i] = fams[b][0].numpy()
        rec['cfr%d_t' % i] = np.float32(tval)
        rec['cfr%d_ft0' % i] = r0[0].numpy()
        rec['cfr%d_ft1' % i] = r1[0].numpy()
        print('cfr%d bit-exact: %s' % (i, bool(torch.equal(r0, m0) and torch.equal(r1, m1))))
    np.savez_compressed(os.path.join(OUT, 'warps_24x40.npz'), **rec)

    # ---------------- kernel-level: FGAC (absolute-coordinate sampling) and pixel_reshuffle ----------------
    H, W = 16, 24
    g = torch.Generator().manual_seed(21)
    ref = torch.tanh(torch.randn(1, 64, H, W, generator=g))
    src = torch.tanh(torch.randn(1, 64, H, W, generator=g))
    rec = dict(ref=ref[0].numpy(), src=src[0].numpy())
    fg = net.FAC_FB_Module.shared_FGAC
    flows = {'inrange': torch.rand(1, 2, H, W, generator=g) * torch.tensor([W - 1.0, H - 1.0]).view(1, 2, 1, 1),
             'mixed': torch.randn(1, 2, H, W, generator=g) * 14,
             'beyond': torch.rand(1, 2, H, W, generator=g) * 10 + torch.tensor([W - 3.0, H - 3.0]).view(1, 2, 1, 1)}
    for name, fl in flows.items():
        with torch.no_grad():
            out, w, _ = fg(ref, src, fl)
            mo, mw = O.fgac(sd, 'FAC_FB_Module.shared_FGAC', ref, src, fl)
        rec['flow_' + name] = fl[0].numpy()
        rec['out_' + name] = out[0].numpy()
        rec['gate_' + name] = w[0].numpy()
        print('fgac/%-8s oracle-vs-reference %.3e' % (name, float((out - mo).abs().max())))
    xs = torch.rand(1, 12, 8, 12, generator=g)
    rec['s2d_in'] = xs[0].numpy()
    rec['s2d_out'] = R.pixel_reshuffle(xs, 2)[0].numpy()
    assert torch.equal(R.pixel_reshuffle(xs, 2), O.space_to_depth(xs, 2))
    np.savez_compressed(os.path.join(OUT, 'fgac_16x24.npz'), **rec)
    print('worst e2e oracle-vs-reference diff %.3e' % worst)


def load_reference_utils():
    """utils.py of the reference as a module: it does not parse as shipped (TabError at
    utils.py:271,273 -> expandtabs) and imports cv2 / skimage / torchvision, absent here.  The two cv2 functions its SSIM
    uses are given their documented formulas (getGaussianKernel: normalised exp(-(i-(k-1)/2)^2 / (2 sigma^2));
    filter2D: correlation, anchor at the kernel centre -- only the [5:-5] "valid" interior is read, so the border mode
    does not matter) -- stated in tests/golden/README as "reference code, stubbed cv2 primitives"."""
    import scipy.ndimage
    cv2 = types.ModuleType('cv2')

    def getGaussianKernel(ksize, sigma):
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)

    def filter2D(img, ddepth, kernel):
        img, kernel = np.asarray(img, np.float64), np.asarray(kernel, np.float64)
        if img.ndim == 3:                                     # cv2 filters every channel of an [h,w,c] image with the 2-D kernel
            kernel = kernel[:, :, None]
        return scipy.ndimage.correlate(img, kernel, mode='mirror')
    cv2.getGaussianKernel, cv2.filter2D = getGaussianKernel, filter2D
    sys.modules['cv2'] = cv2
    for name in ['skimage', 'skimage.metrics', 'torchvision', 'torchvision.models', 'torchvision.transforms']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['skimage.metrics'].structural_similarity = None
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    U = types.ModuleType('utils_ref')
    exec(compile(open('/root/reference/utils.py').read().expandtabs(8), 'utils_ref', 'exec'), U.__dict__)
    return U


def round2():
    """Fixtures added in round 2 (the round-1 files above are not rewritten): config 1 of BASELINE.json, the uint8
    conversions of the loader / writer, PSNR + SSIM of the reference's evaluation."""
    sd = synthetic_state_dict(0)
    net = R.DeMFInet(ARGS).eval()
    net.load_state_dict(sd)
    U = load_reference_utils()
    # ---- config 1: 256x256, N_tst = 1, x2 (t = 0.5), CPU fp32 ------------------------------
    H = W = 256
    x = synthetic_window(H, W, 1)
    t = torch.tensor([[0.5]], dtype=torch.float32)
    with torch.no_grad():
        d1, fin, flows, occs, ov = net(x, t, 1)
        mine = O.forward(sd, x, t, 1)
    gt = x[0, :, 0].numpy()                                  # fixed pseudo ground truth (B0)
    u8 = lambda z: np.around(U.denorm255_np(z.numpy().astype(np.float64))).astype(np.uint8)
    rec = dict(H=H, W=W, seed=1, t=np.float32(0.5), N=1, weight_seed=0,
               St=fin[0][2][0].numpy(), flows_last=flows[-1][0].numpy(), occ_last=occs[-1][0].numpy(),
               finals_u8=np.stack([u8(z[0]) for z in fin[0]]), d1_u8=np.stack([u8(z[0]) for z in d1]),
               psnr_St_vs_B0=np.float64(U.psnr(np.around(U.denorm255_np(fin[0][2][0].numpy().astype(np.float64))),
                                               np.around(U.denorm255_np(gt.astype(np.float64))))))
    np.savez_compressed(os.path.join(OUT, 'cfg1_256x256_t0500_n1.npz'), **rec)
    print('cfg1_256x256           oracle-vs-reference max|diff| = %.3e   psnr(St,B0) = %.4f dB' %
          (max(float((a - b).abs().max()) for a, b in zip(fin[0], mine[1][0])), rec['psnr_St_vs_B0']))

    # ---- uint8 conversions: loader (utils.py:224-238) and writer (utils.py:718-721 + main.py:1165-1178) ---------
    g = np.random.RandomState(7)
    frames = g.randint(0, 256, size=(4, 20, 28, 3)).astype(np.uint8)
    frames[0, 0, :8, 0] = [0, 1, 2, 127, 128, 253, 254, 255]           # the end points and the middle, explicitly
    ten = U.RGBframes_np2Tensor(frames, 3)                        # [3,T,h,w] fp32
    assert torch.equal(ten, O.frames_u8_to_tensor(list(frames)))
    # writer input: fp32 frames around every rounding boundary k/255 and outside [-1,1]
    k = np.arange(0, 256, dtype=np.float64)
    edge = (k / 255.0) * 2 - 1
    vals = np.concatenate([edge, np.nextafter(edge, -9), np.nextafter(edge, 9), edge + 1e-4, edge - 1e-4,
                           [-1.5, 1.5, -1.0000001, 1.0000001, 0.0, -0.0]]).astype(np.float32)
    pred = np.resize(np.concatenate([vals, g.uniform(-1.1, 1.1, 3 * 20 * 28 - vals.size).astype(np.float32)]),
                     (3, 20, 28)).astype(np.float32)
    out_s = np.transpose(np.squeeze(U.denorm255_np(pred.astype(np.float64)[None])), [1, 2, 0]).astype(np.uint8)   # S0/S1 path
    out_t = U.denorm255_np(np.transpose(pred.astype(np.float64), [1, 2, 0])[:, :, ::-1]).astype(np.uint8)[:, :, ::-1]   # St path
    assert np.array_equal(out_s, out_t) and np.array_equal(out_s, O.frame_to_u8(pred))
    np.savez_compressed(os.path.join(OUT, 'u8io_20x28.npz'), frames=frames, tensor=ten.numpy(), pred=pred, out_u8=out_s)
    print('u8io                   loader / writer oracle helpers == reference: True')

    # ---- evaluation: psnr (utils.py:652-660), ssim (utils.py:663-705) on np.around(denorm255_np(.)) frames --------
    a = fin[0][2][0].numpy()[:, 40:136, 60:188]
    b = np.clip(a + g.normal(0, 0.05, a.shape).astype(np.float32), -1.2, 1.2)
    ia = np.around(U.denorm255_np(np.transpose(a.astype(np.float64), [1, 2, 0])))
    ib = np.around(U.denorm255_np(np.transpose(b.astype(np.float64), [1, 2, 0])))
    np.savez_compressed(os.path.join(OUT, 'metrics_96x128.npz'), a=a, b=b, psnr=np.float64(U.psnr(ia, ib)),
                        ssim=np.float64(U.ssim(ia, ib)), psnr_same=np.float64(U.psnr(ia, ia)), ssim_same=np.float64(U.ssim(ia, ia)))
    print('metrics                psnr %.6f dB  ssim %.8f' % (U.psnr(ia, ib), U.ssim(ia, ib)))


def patched_fgac(net, rr, sr):
    """The reference's FGAC.forward with its two hard-coded radii (DeMFInet.py:401-402) overridden IN MEMORY: the source
    text of the method is fetched with inspect, the two constant assignments are rewritten, and the result is bound to the
    model's own FGAC module (its weights).  Nothing is written to disk; the generalised code path (403-445) is the
    reference's own."""
    import inspect
    import textwrap
    src = textwrap.dedent(inspect.getsource(R.FGAC.forward))
    assert src.count('rr = 0') == 1 and src.count('sr = 0') == 1
    src = src.replace('rr = 0', 'rr = %d' % rr).replace('sr = 0', 'sr = %d' % sr)
    ns = {}
    exec(compile(src, 'FGAC_forward_patched', 'exec'), R.__dict__, ns)
    fg = net.FAC_FB_Module.shared_FGAC
    return lambda ref, source, flow: ns['forward'](fg, ref, source, flow)


def fgac_window_fixtures():
    """Generalised FGAC (rr, sr > 0): outputs of the PATCHED reference (see patched_fgac) -> tests/golden/fgac_window_16x24.npz."""
    sd = synthetic_state_dict(0)
    net = R.DeMFInet(ARGS).eval()
    net.load_state_dict(sd)
    H, W = 16, 24
    g = torch.Generator().manual_seed(31)
    ref = torch.tanh(torch.randn(1, 64, H, W, generator=g))
    src = torch.tanh(torch.randn(1, 64, H, W, generator=g))
    flows = {'inrange': torch.rand(1, 2, H, W, generator=g) * torch.tensor([W - 1.0, H - 1.0]).view(1, 2, 1, 1),
             'mixed': torch.randn(1, 2, H, W, generator=g) * 9}
    rec = dict(ref=ref[0].numpy(), src=src[0].numpy())
    fg = net.FAC_FB_Module.shared_FGAC
    for (rr, sr) in ((1, 0), (2, 0), (1, 1)):
        fwd = patched_fgac(net, rr, sr)
        for name, fl in flows.items():
            grabbed = {}
            hook = fg.fusion.register_forward_pre_hook(lambda mod, inp: grabbed.__setitem__('fac', inp[0].detach().clone()))
            with torch.no_grad():
                out, w, _ = fwd(ref, src, fl)
                mo, mw, mfac, matt = O.fgac_general(sd, 'FAC_FB_Module.shared_FGAC', ref, src, fl, rr, sr, 0)
            hook.remove()
            tag = 'rr%d_sr%d_%s' % (rr, sr, name)
            rec['flow_' + name] = fl[0].numpy()
            rec['fac_' + tag] = grabbed['fac'][0].numpy()
            rec['out_' + tag] = out[0].numpy()
            rec['gate_' + tag] = w[0].numpy()
            print('fgac_window %-22s oracle-vs-patched-reference: FAC %.3e  out %.3e' %
                  (tag, float((grabbed['fac'] - mfac).abs().max()), float((out - mo).abs().max())))
    np.savez_compressed(os.path.join(OUT, 'fgac_window_16x24.npz'), **rec)


def round6():
    """Fixtures added in round 6: (a) the visualisation / training return tuples of the UNPATCHED reference (DeMFInet.py:167-176,
    454-496), (b) a second weight regime -- small flows, unsaturated occlusion (synthetic_state_dict(flow_gain=0.3)) -- end to end."""
    x = synthetic_window(64, 96, 1)
    t = torch.tensor([[0.5]], dtype=torch.float32)
    sd = synthetic_state_dict(0)
    viz_args = types.SimpleNamespace(**{**vars(ARGS), 'visualization_flag': True})
    nv = R.DeMFInet(viz_args).eval()
    nv.load_state_dict(sd)
    with torch.no_grad():
        out_v = nv(x, t, 1)                                    # 7-tuple of DeMFInet.py:174-176
        out_t = R.DeMFInet(ARGS).eval()
        out_t.load_state_dict(sd)
        out_t = out_t(x, t, 1, True)                           # is_training: 7-tuple of 170-172 (eval-mode numbers)
        mine_bw, mine_diff = O.forward_extras(sd, x)
    bw, diffs = out_v[5], out_v[6]
    assert len(out_v) == 7 and len(bw) == 5 and len(diffs) == 4 and len(out_t) == 7
    assert all(torch.equal(a, b) for a, b in zip(bw[0], bw[2])) and torch.equal(diffs[0], diffs[2]) and torch.equal(diffs[1], diffs[3])
    rec = dict(H=64, W=96, seed=1, t=np.float32(0.5), N=1, weight_seed=0,
               bw=torch.stack([torch.stack([m[0, 0] for m in bw[b]]) for b in range(2)]).numpy(),          # [2, 6, H, W]
               diff=torch.stack([diffs[b][0, 0] for b in range(2)]).numpy(),                                # [2, H, W]
               flow_01=bw[4][0][0].numpy(), flow_10=bw[4][1][0].numpy(),
               St=out_v[1][0][2][0].numpy(),
               train_diff=torch.stack([out_t[5][b][0, 0] for b in range(2)]).numpy(),
               train_rflow=torch.stack([out_t[6][0][i][0] for i in range(2)]).numpy())                       # [2, 2, H, W]
    np.savez_compressed(os.path.join(OUT, 'extras_64x96_t0500_n1.npz'), **rec)
    d = max(float((a - b).abs().max()) for b in range(2) for a, b in zip(bw[b], mine_bw[b]))
    d = max(d, max(float((a - b).abs().max()) for a, b in zip(diffs, mine_diff)))
    print('extras_64x96           oracle-vs-reference max|diff| = %.3e (gates, normalised maps, diff)' % d)

    sd2 = synthetic_state_dict(0, flow_gain=0.3)
    net = R.DeMFInet(ARGS).eval()
    net.load_state_dict(sd2)
    for tag, (H, W, seed, tval, N) in {'e2e_smallflow_64x96_t0500_n3': (64, 96, 1, 0.5, 3), 'e2e_smallflow_64x96_t0125_n2': (64, 96, 2, 0.125, 2)}.items():
        x = synthetic_window(H, W, seed)
        t = torch.tensor([[tval]], dtype=torch.float32)
        with torch.no_grad():
            d1, fin, flows, occs, ov = net(x, t, N)
            mine = O.forward(sd2, x, t, N)
        rec = dict(H=H, W=W, seed=seed, t=np.float32(tval), N=N, weight_seed=0, flow_gain=np.float32(0.3),
                   d1=torch.stack([z[0] for z in d1]).numpy(),
                   finals=torch.stack([torch.stack([z[0] for z in f]) for f in fin]).numpy(),
                   flows=torch.stack([z[0] for z in flows]).numpy(),
                   occs=torch.stack([z[0] for z in occs]).numpy(),
                   overlay=ov[0].numpy())
        np.savez_compressed(os.path.join(OUT, tag + '.npz'), **rec)
        d = max(float((a - b).abs().max()) for a, b in zip(fin[-1], mine[1][-1]))
        print('%-30s oracle-vs-reference %.3e   |flow| max %.2f px, mean %.2f px; occlusion in [%.3f, %.3f], saturated (<0.02 or >0.98): %.1f %%' %
              (tag, d, float(flows[-1].abs().max()), float(flows[-1].abs().mean()), float(occs[-1].min()), float(occs[-1].max()),
               100.0 * float(((occs[-1] < 0.02) | (occs[-1] > 0.98)).float().mean())))
    # for the record: the same statistics of the first regime
    net.load_state_dict(sd)
    with torch.no_grad():
        _, _, flows, occs, _ = net(synthetic_window(64, 96, 1), torch.tensor([[0.5]]), 3)
    print('%-30s (regime 1)            |flow| max %.2f px, mean %.2f px; occlusion in [%.3f, %.3f], saturated: %.1f %%' %
          ('e2e_64x96_t0500_n3', float(flows[-1].abs().max()), float(flows[-1].abs().mean()), float(occs[-1].min()), float(occs[-1].max()),
           100.0 * float(((occs[-1] < 0.02) | (occs[-1] > 0.98)).float().mean())))


if __name__ == '__main__':
    if '--round6' in sys.argv:
        round6()
        sys.exit(0)
    if '--fgac-window' in sys.argv:
        fgac_window_fixtures()
        sys.exit(0)
    if '--round2' in sys.argv:
        round2()
    else:
        main()
        round2()
        round6()
