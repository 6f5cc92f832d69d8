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


if __name__ == '__main__':
    main()
