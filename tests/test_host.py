"""CPU tests of the host side: state_dict contract, C-ABI exports, weight repack, and the launch plan
interpreted on CPU (tests/plan_sim.py) against the oracle.  No kernel is launched here."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from demfi_amd import _lib as L
from demfi_amd.engine import Engine, Plan, _Dst
from demfi_amd.model import DeMFInet
from demfi_amd.spec import HyperParams, state_dict_shapes
from demfi_amd.weights import synthetic_state_dict, synthetic_window
from oracle import demfi_oracle as O
from tests.plan_sim import PlanSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract():
    shapes = state_dict_shapes()
    assert len(shapes) == 260
    assert sum(int(np.prod(s)) for s in shapes.values()) == 7408284
    m = DeMFInet(HyperParams())
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert tuple(sd['Decoder_res.3.conv1.weight'].shape) == (64, 64, 1, 3, 3)
    assert tuple(sd['Booster_Module.GB.convq2.weight'].shape) == (64, 128, 5, 1)
    m.load_state_dict(synthetic_state_dict(0))                                 # same keys -> strict load works
    assert len(state_dict_shapes(HyperParams(shared_FGAC_flag=False))) == 270
    assert sum(int(np.prod(s)) for s in state_dict_shapes(HyperParams(shared_FGAC_flag=False)).values()) == 7495133


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'demfi_hip.h')).read()
    body = hdr[hdr.index('---- library / device'):]
    declared = set(re.findall(r'\b(demfi_[a-z0-9_]+)\s*\(', body))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.demfi_abi_version() == L.ABI_VERSION == 8
    assert C.sizeof(L.Batch) == 8 + 4 * 8 + 32 * 8
    assert C.sizeof(L.View) == 48 and C.sizeof(L.Piece) == 64 and C.sizeof(L.Chunk) == 24 and C.sizeof(L.Seg) == 168


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    m = DeMFInet(HyperParams())
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 4, 32, 32), torch.tensor([[0.5]]), 1)
    with pytest.raises(RuntimeError):                         # round 6: the training / visualisation tuples exist, and are HIP-only too
        m(torch.zeros(1, 3, 4, 32, 32), torch.tensor([[0.5]]), 1, is_training=True)


def test_pack_rejects_bad_arguments():
    lib = L.load()
    n = C.c_int64(0)
    w = np.zeros((4, 4, 1, 1), np.float32)
    cin = np.arange(16, dtype=np.int32)          # indices 4..15 exceed cin=4 -> rejected when packing
    nks = np.asarray([1], np.int32)
    cout = np.arange(32, dtype=np.int32)
    cout[4:] = -1
    st = lib.demfi_pack_conv_weights(w.ctypes.data, 4, 4, 1, 1, cin.ctypes.data, 16, nks.ctypes.data, 1, cout.ctypes.data,
                                     32, 1, L.F16, None, C.byref(n))
    assert st == 0 and n.value == 64 * 16
    out = np.zeros(n.value, np.uint8)
    st = lib.demfi_pack_conv_weights(w.ctypes.data, 4, 4, 1, 1, cin.ctypes.data, 16, nks.ctypes.data, 1, cout.ctypes.data,
                                     32, 1, L.F16, out.ctypes.data, C.byref(n))
    assert st == -1 and b'cin_map' in lib.demfi_last_error()
    st = lib.demfi_pack_conv_weights(w.ctypes.data, 4, 4, 1, 1, cin.ctypes.data, 16, nks.ctypes.data, 1, cout.ctypes.data,
                                     48, 2, L.F16, None, C.byref(n))
    assert st == -1


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_weight_pack_roundtrip_and_conv_sim(dtype):
    """demfi_pack_conv_weights -> unpack (test-side inverse) reproduces W[cout_map][cin_map]; odd channel counts,
    mixed fat/thin inputs, split outputs."""
    torch.manual_seed(0)
    H, W = 8, 32
    pl = Plan(H, W, dtype, 'cpu')
    a = pl._fat(H, W, 24)
    b = pl._thin(5)
    a.copy_(torch.randn(a.shape))
    b.copy_(torch.randn(b.shape))
    o1 = pl._fat(H, W, 16)
    o2 = pl._thin(3)
    wt = torch.randn(19, 29, 3, 3) * 0.1
    bs = torch.randn(19)
    seg = []
    # logical input = [thin(0..4) | fat(5..28)], outputs: couts 3..18 -> fat, 0..2 -> thin
    pl.conv(seg, 'case', [pl.tsrc(b, range(0, 5)), pl.fsrc(a, 5)],
            [_Dst(pl.fview(o1), range(3, 19), L.ACT_RELU), _Dst(pl.tview(o2), range(0, 3))], H, W, weight=wt, bias=bs)
    pl._upload()
    sim = PlanSim(pl)
    sim.conv(pl._descs[0])
    xin = torch.cat([b, a[0].permute(2, 0, 1).float()], 0)[None]
    if dtype == torch.float16:
        xin = xin.half().float()
        wt = wt.half().float()
    ref = torch.nn.functional.conv2d(xin, wt, bs, padding=1)[0]
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert (o2 - ref[0:3]).abs().max() < tol
    assert (o1[0].permute(2, 0, 1).float() - torch.relu(ref[3:19])).abs().max() < (tol if dtype == torch.float32 else 2e-2)


def test_plan_matches_oracle_fp32(synthetic_sd):
    """The whole launch plan (engine.py) interpreted on CPU == oracle forward, fp32, N=2."""
    H, W, N = 32, 64, 2
    eng = Engine(synthetic_sd, H, W, torch.float32, 'cpu', max_updates=N)
    assert eng.n_convs == 74 + 30 + 25 * N
    x = synthetic_window(H, W, 4)
    PlanSim(eng).forward(x, 0.375, N)
    with torch.no_grad():
        d1, fin, flows, occs, ov = O.forward(synthetic_sd, x, torch.tensor([[0.375]]), N)
    for i in range(3):
        assert (eng.sharp1[3 * i:3 * i + 3] - d1[i][0]).abs().max() < 1e-4
        for it in range(N):
            assert (eng.finals[it, i] - fin[it][i][0]).abs().max() < 1e-4
    for i in range(N + 1):
        assert (eng.delta[i, 0:4] - flows[i][0]).abs().max() < 2e-4
        assert (eng.occ[i:i + 1] - occs[i][0]).abs().max() < 1e-4
    assert torch.equal(eng.overlay, ov[0])


def test_plan_with_visualisation_extras(synthetic_sd):
    """DEMFI_HP_EXTRAS: the plan also computes FGAC's gates, min-max normalised channel-mean maps and diff (DeMFInet.py:454-496) -- the
    interpreted plan against the oracle's restatement, fp32; the frames do not change."""
    H, W, N = 32, 64, 1
    hp = HyperParams(visualization_flag=True)
    eng = Engine(synthetic_sd, H, W, torch.float32, 'cpu', max_updates=N, hp=hp)
    plain = Engine(synthetic_sd, H, W, torch.float32, 'cpu', max_updates=N)
    assert eng.n_launches(N)[0] == plain.n_launches(N)[0] + 2 * (1 + 2 * 5)     # per direction: 1 - w, five (mean, normalise) pairs
    x = synthetic_window(H, W, 4)
    PlanSim(eng).forward(x, 0.375, N)
    PlanSim(plain).forward(x, 0.375, N)
    assert torch.equal(eng.finals, plain.finals)
    with torch.no_grad():
        bw, diffs = O.forward_extras(synthetic_sd, x)
    for b in range(2):
        assert (eng.gate[b] - bw[b][0][0, 0]).abs().max() < 1e-5
        for k in range(5):
            assert (eng.viz[b, k] - bw[b][k + 1][0, 0]).abs().max() < 2e-5, (b, k)
        assert (eng.viz[b, 5] - diffs[b][0, 0]).abs().max() < 2e-5


def test_plan_non_shared_fgac():
    hp = HyperParams(shared_FGAC_flag=False)
    sd = synthetic_state_dict(3, hp)
    H, W = 32, 32
    eng = Engine(sd, H, W, torch.float32, 'cpu', max_updates=1, hp=hp)
    x = synthetic_window(H, W, 6)
    PlanSim(eng).forward(x, 0.5, 1)
    with torch.no_grad():
        out = O.forward(sd, x, torch.tensor([[0.5]]), 1, shared_fgac=False)
    assert (eng.finals[0, 2] - out[1][0][2][0]).abs().max() < 1e-4


@pytest.mark.parametrize('rr,sr,fmap', [(1, 0, 0), (1, 1, 1)])
def test_plan_generalised_fgac(rr, sr, fmap):
    """hp.fgac_rr / fgac_sr > 0: the plan gains conv_source_k, the optional pooling and the window op (both path dtypes since
    round 3); interpreted on CPU it follows the oracle's generalised FGAC (itself pinned to the patched reference)."""
    hp = HyperParams(fgac_rr=rr, fgac_sr=sr, fgac_map=fmap)
    sd = synthetic_state_dict(0)
    H, W = 32, 32
    e32 = Engine(sd, H, W, torch.float32, 'cpu', max_updates=1, hp=hp)          # fp32 instantiation of the window kernel (round 3)
    assert [op.kind for op in e32.ops(0)].count(8) == 2
    eng = Engine(sd, H, W, torch.float16, 'cpu', max_updates=1, hp=hp)
    kinds = [op.kind for op in eng.ops(0)]
    assert kinds.count(8) == 2 and kinds.count(4) == 0 and kinds.count(9) == (4 if sr else 0)
    x = synthetic_window(H, W, 6)
    PlanSim(eng).forward(x, 0.5, 1)
    with torch.no_grad():
        ref = O.forward(sd, x, torch.tensor([[0.5]]), 1, fgac_radii=(rr, sr, fmap))
        base = O.forward(sd, x, torch.tensor([[0.5]]), 1)
    got = eng.finals[0, 2].float().numpy()
    assert O.psnr(got, ref[1][0][2][0].numpy()) > 38.0                           # fp16 storage vs fp32 oracle
    assert O.psnr(got, ref[1][0][2][0].numpy()) > O.psnr(got, base[1][0][2][0].numpy()) + 3.0   # and it is NOT the rr = 0 result


def test_fp16_plan_with_hoisted_partial_convs(synthetic_sd):
    """The fp16 plan splits two layers by linearity (enc1: t-independent aF half hoisted into the trunk; Dec_first_2:
    recursion-invariant planes hoisted, the rest on the persistent kernels).  Interpreted on CPU it must still be the
    network: compare with the fp32 oracle (fp16 storage only costs ~50 dB) and count the extra launches."""
    H, W, N = 32, 64, 2
    eng = Engine(synthetic_sd, H, W, torch.float16, 'cpu', max_updates=N)
    names = [op.name.decode() for op in eng.ops(0)] + [op.name.decode() for op in eng.ops(1)] + [op.name.decode() for op in eng.ops(2, 0)]
    assert 'Refine_Module.enc1#aF' in names and 'Refine_Module.enc1#t' in names and 'Refine_Module.enc1' not in names
    assert {'Dec_first_2#win', 'Dec_first_2#dyn', 'Dec_first_2#rec', 'Booster_Module.Mixer.conv_ref1#win'} <= set(names)
    assert 'Dec_first_2' not in names and 'Booster_Module.Mixer.conv_ref1' not in names
    x = synthetic_window(H, W, 4)
    PlanSim(eng).forward(x, 0.375, N)
    with torch.no_grad():
        ref = O.forward(synthetic_sd, x, torch.tensor([[0.375]]), N)
    for i in range(3):
        assert O.psnr(eng.finals[N - 1, i].float().numpy(), ref[1][N - 1][i][0].numpy()) > 42.0
    assert (eng.delta[N, 0:4].float() - ref[2][N][0]).abs().median() < 2e-2


def test_ch_reducer_descriptor_is_the_streamed_weight_kernels(synthetic_sd):
    """Ch_Reducer (7x7, 192 -> 64) runs on conv_wstream_c64_kernel (conv.hip) only if the plan hands it the shape that kernel owns:
    fp16, six single-piece 32-channel chunks (64-byte records: the two halves of the three 64-channel images of rF, same strides), 64
    packed couts in the permuted order,
    one NHWC destination without residual.  A builder change that made the layer ineligible would silently fall back to the general
    kernel (a 25 % slower layer, invisible to the parity tests): pin the descriptor here, per-t and batched."""
    from demfi_amd.engine import SEG_HEAD, SEG_TB_HEAD
    eng = Engine(synthetic_sd, 32, 64, torch.float16, 'cpu', max_updates=1, n_ctx=2)
    for seg in (SEG_HEAD, SEG_TB_HEAD):
        op = [o for o in eng.ops(seg) if o.name.decode() == 'Ch_Reducer'][0]
        d = eng.conv_desc(op.conv)
        assert (d.kh, d.kw, d.stride, d.pad_y, d.pad_x) == (7, 7, 1, 3, 3)
        assert d.cout_perm == 1 and d.rec_bytes == 64 and d.n_chunks == 6 and d.cout_pad == 64 and d.nco == 2
        assert d.batch == (2 if seg == SEG_TB_HEAD else 1)
        first = d.pieces[d.chunks[0].first_piece].v
        for c in range(6):
            ch = d.chunks[c]
            pc = d.pieces[ch.first_piece]
            assert ch.n_pieces == 1 and ch.nks == 2 and pc.nch == 32 and pc.fat == 1 and pc.up_shift == 0
            assert (pc.v.sx, pc.v.sy, pc.v.sc, pc.v.sb) == (64, 64 * 64, 1, first.sb)
            assert pc.v.ptr == first.ptr + (c // 2) * 32 * 64 * 64 * 2 + (c % 2) * 64          # image c // 2 of rF, channel half c % 2
        sg = d.segs[d.sub_seg[0]]
        assert d.sub_seg[1] == d.sub_seg[0] and not sg.res.ptr and sg.dst.sc == 1 and not sg.dst.is_f32
    # the fp32 plan keeps the layer on the general kernel (no permuted cout order)
    e32 = Engine(synthetic_sd, 32, 64, torch.float32, 'cpu', max_updates=1)
    op = [o for o in e32.ops(SEG_HEAD) if o.name.decode() == 'Ch_Reducer'][0]
    assert e32.conv_desc(op.conv).cout_perm == 0


def test_unet_layers_carry_the_shape_of_the_phase_kernel(synthetic_sd):
    """Round 6: the 4x4 stride-2 encoders and the >= 96-channel 3x3 layers of the refinement UNet (DeMFInet.py:575-603) and FGAC's w_gen
    run on wsconv.hip only if the plan hands them the shape that kernel owns -- 64-byte records, one 32-channel NHWC piece (or a
    16-channel tail + zero padding) per chunk, two 32-cout subtiles per work item, permuted cout order.  A builder change that made them
    ineligible would silently put them back on the general kernel (2-3x slower, invisible to the parity tests): pin the descriptors."""
    from demfi_amd.engine import SEG_HEAD, SEG_TB_HEAD, SEG_TRUNK
    eng = Engine(synthetic_sd, 64, 96, torch.float16, 'cpu', max_updates=1, n_ctx=2)
    want = {'Refine_Module.enc1#aF': (4, 2, 4, 64), 'Refine_Module.enc1#t': (4, 2, 3, 64), 'Refine_Module.enc2': (4, 2, 2, 128),
            'Refine_Module.enc3': (4, 2, 4, 256), 'Refine_Module.dec0': (3, 1, 8, 256), 'Refine_Module.dec1': (3, 1, 12, 128),
            'Refine_Module.dec2': (3, 1, 6, 64), 'FAC_FB_Module.shared_FGAC.w_gen': (3, 1, 4, 64)}
    seen = set()
    for seg in (SEG_TRUNK, SEG_HEAD, SEG_TB_HEAD):
        for op in eng.ops(seg):
            n = op.name.decode()
            if n not in want or op.kind != 0:
                continue
            d = eng.conv_desc(op.conv)
            k, stride, n_chunks, cout_pad = want[n]
            assert (d.kh, d.kw, d.stride, d.n_chunks, d.cout_pad) == (k, k, stride, n_chunks, cout_pad), n
            assert d.rec_bytes == 64 and d.nco == 2 and d.cout_perm == 1, n
            for c in range(d.n_chunks):
                ch = d.chunks[c]
                pc = d.pieces[ch.first_piece]
                assert ch.nks == 2 and pc.fat == 1 and pc.lds_ch == 0 and ((ch.n_pieces == 1 and pc.nch == 32) or (ch.n_pieces == 2 and pc.nch == 16)), (n, c)
            seen.add(n)
    assert seen == set(want)
    up = [eng.conv_desc(o.conv) for o in eng.ops(SEG_HEAD) if o.name.decode() == 'Refine_Module.dec1'][0]
    assert [up.pieces[up.chunks[c].first_piece].up_shift for c in range(12)] == [1] * 8 + [0] * 4      # cat[up(d0), u2]
    # the fp32 plan keeps them on the general kernel
    e32 = Engine(synthetic_sd, 64, 96, torch.float32, 'cpu', max_updates=1)
    op = [o for o in e32.ops(SEG_HEAD) if o.name.decode() == 'Refine_Module.enc2'][0]
    assert e32.conv_desc(op.conv).cout_perm == 0


def test_fused_dec_first_2_experiment_is_still_the_network(synthetic_sd):
    """DEMFI_DF2_FUSE=1 (round-6 experiment, measured neutral, not the product): Dec_first_2's per-recursion part as ONE launch whose third
    unit is [16 channels of ref16 through a channel map | the recursion's 8-channel record | 0].  The switch is read once per process:
    a child interprets that plan on the CPU against the oracle."""
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from demfi_amd.engine import Engine\n"
        "from demfi_amd.weights import synthetic_state_dict, synthetic_window\n"
        "from tests.plan_sim import PlanSim\n"
        "from oracle import demfi_oracle as O\n"
        "sd = synthetic_state_dict(0)\n"
        "eng = Engine(sd, 32, 64, torch.float16, 'cpu', max_updates=2)\n"
        "names = [op.name.decode() for op in eng.ops(2, 0)]\n"
        "assert 'Dec_first_2#t' in names and 'Dec_first_2#rec' not in names, names\n"
        "x = synthetic_window(32, 64, 4)\n"
        "PlanSim(eng).forward(x, 0.375, 2)\n"
        "ref = O.forward(sd, x, torch.tensor([[0.375]]), 2)\n"
        "for i in range(3):\n"
        "    assert O.psnr(eng.finals[1, i].float().numpy(), ref[1][1][i][0].numpy()) > 42.0\n"
        "print('ok')\n")
    e = dict(os.environ, DEMFI_DF2_FUSE='1')
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_batched_per_t_plan_equals_per_context_plans(synthetic_sd, dtype):
    """demfi_forward_tb: ONE op list for all per-t contexts (convolutions batched over the contexts through the contiguous
    copies of every per-t buffer, point-wise ops once per context).  Interpreted on CPU it must leave in every context exactly
    what that context's own op lists leave there -- also for the batch-3 D1 layers, the window-level residuals (batch stride
    0) and the pinned-frame pieces of Ch_Reducer."""
    H, W, N, NC = 32, 64, 2, 3
    eng = Engine(synthetic_sd, H, W, dtype, 'cpu', max_updates=N, n_ctx=NC)
    x = synthetic_window(H, W, 6)
    ts = [0.25, 0.5, 0.875]
    sim = PlanSim(eng)
    want = []
    for c, t in enumerate(ts):
        eng.use_ctx(c)
        sim.forward(x, t, N)
        want.append({k: eng._ctxs[0][c][k].clone() for k in ('finals', 'delta', 'occ', 'sharp1')})
        for k in ('finals', 'delta', 'occ', 'sharp1'):
            eng._ctxs[0][c][k].zero_()
    eng.use_ctx(0)
    sim.forward_tb(x, ts, N)
    for c in range(NC):
        for k, v in want[c].items():
            assert torch.equal(eng._ctxs[0][c][k], v), (c, k)
    assert not torch.equal(want[0]['finals'], want[1]['finals'])
    # one launch per convolution for all contexts (batch dimension); CFR and the thin warps also ONE launch (demfi_batch, ABI v5:
    # pointers of context 0 + one byte stride per pointer, 0 for window-level buffers); fat warp and plane packs NC launches
    from demfi_amd.engine import SEG_HEAD, SEG_TB_HEAD, SEG_TB_ITER
    one, tb = eng.ops(SEG_HEAD), eng.ops(SEG_TB_HEAD)
    # convolutions, fused residual blocks, CFR, the thin warps and (round 5) the fat warps and the plane packs: ONE launch each for
    # all contexts -- the batched per-t sequence has exactly the launches of ONE context's sequence
    assert len(tb) == len(one)
    packs = [o for o in tb if o.kind == 1]
    assert packs and all(o.bt.nb == NC and o.bt.o > 0 for o in packs)
    fat = [o for o in tb if o.kind == 7 and o.nch != 3]
    assert len(fat) == 2 and all(o.bt.nb == NC and o.bt._pad == 1 and o.bt.o > 0 for o in fat)      # one grid slice per context
    assert fat[0].bt.a == 0 and fat[1].bt.a > 0           # Ft warps the window's trunk features (shared), rF this context's refined ones
    cfr = [o for o in tb if o.kind == 6]
    assert len(cfr) == 1 and cfr[0].bt.nb == NC and cfr[0].bt.p[0] == 0 and cfr[0].bt.p[3] > 0 and cfr[0].bt.t > 0   # flows shared, outputs strided
    thin = [o for o in eng.ops(SEG_TB_ITER, it=0) if o.kind == 7]
    assert len(thin) == 1 and thin[0].nch == 3 and thin[0].bt.nb == NC and thin[0].bt.o > 0
    d1 = [o for o in tb if o.name.decode() == 'Dec_first'][0]
    assert eng.conv_desc(d1.conv).batch == 3 * NC


def test_load_checkpoint_reads_the_reference_layout(tmp_path, synthetic_sd):
    """main.py:316, 351: ``model_net.load_state_dict(checkpoint['state_dict_Model'])`` on the file SaveManager wrote
    (utils.py:47-66 stores the state_dict beside epoch / optimizer entries).  ``--checkpoint`` of bench.py / demfi_amd.clip goes
    through weights.load_checkpoint; DataParallel's ``module.`` prefix and a bare state_dict file are accepted, junk is not."""
    from demfi_amd.weights import load_checkpoint
    sd = {k: v.clone() for k, v in synthetic_sd.items()}
    p1, p2, p3 = str(tmp_path / 'a_latest.pt'), str(tmp_path / 'b.pt'), str(tmp_path / 'c.pt')
    torch.save({'last_epoch': 7, 'state_dict_Model': {'module.' + k: v.half() for k, v in sd.items()}, 'best_PSNR': 0.0}, p1)
    torch.save(sd, p2)
    torch.save({'state_dict_Model': {}}, p3)
    got = load_checkpoint(p1)
    assert set(got) == set(sd) and all(v.dtype == torch.float32 for v in got.values())
    assert all(torch.equal(got[k], sd[k].half().float()) for k in sd)
    m = DeMFInet(HyperParams())
    m.load_state_dict(load_checkpoint(p2))                      # strict: 260 keys, shapes as registered
    assert torch.equal(m.state_dict()['Dec_last2_2.weight'], sd['Dec_last2_2.weight'])
    with pytest.raises(ValueError):
        load_checkpoint(p3)
    # a genuine reference file carries numpy scalars beside the state_dict (main.py:262-271: best / last PSNR, SSIM and
    # the loss meters as numpy.float64); torch >= 2.6 loads with weights_only=True, which must still accept them
    import numpy as np
    p4 = str(tmp_path / 'd.pt')
    torch.save({'last_epoch': 3, 'state_dict_Model': sd, 'intp_testSSIM': np.float64(0.93), 'deblur_testSSIM': np.float64(0.95),
                'best_PSNR': np.float64(31.2), 'loss_meter': torch.tensor(0.01), 'intp_testPSNR': np.float32(30.0)}, p4)
    got4 = load_checkpoint(p4)
    assert set(got4) == set(sd) and torch.equal(got4['Dec_last2_2.weight'], sd['Dec_last2_2.weight'])


def test_bench_self_launches_under_torch_distributed_run(monkeypatch):
    """VERDICT r4 missing #1: a bare ``python bench.py --gpus N`` (N > 1, no WORLD_SIZE) re-executes itself as one rank per GPU under
    torch.distributed.run on 127.0.0.1; the caller's flags pass through unchanged; the launcher's exit status is the run's."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    argv = b.self_launch_argv(8, ['--gpus', '8', '--steps', '4', '--warmup', '1'], port=29511)
    assert argv[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert argv[argv.index('--nproc-per-node') + 1] == '8' and argv[argv.index('--master-addr') + 1] == '127.0.0.1'
    assert argv[argv.index('--master-port') + 1] == '29511'
    assert os.path.samefile(argv[-7], os.path.join(root, 'bench.py')) and argv[-6:] == ['--gpus', '8', '--steps', '4', '--warmup', '1']
    free = b.self_launch_argv(2, [])
    assert 1024 < int(free[free.index('--master-port') + 1]) < 65536
    # main() takes the self-launch branch exactly when WORLD_SIZE is absent and --gpus > 1, and returns the child's status
    calls = []
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(b, 'self_launch', lambda n: calls.append(n) or 3)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert calls == [2] and ei.value.code == 3


def test_rank_affinity_slices_a_numa_node_between_its_ranks():
    """tools/run_node.sh: ranks whose GPUs share a NUMA node get disjoint slices of that node's cores, the node comes
    from the PCI address of HIP device LOCAL_RANK; without NUMA information the cores are split evenly by rank."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('rank_affinity', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'rank_affinity.py'))
    ra = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ra)
    assert ra.parse_cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11] and ra.fmt_cpulist([4, 5]) == '4,5'
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    node_cpus = {0: list(range(0, 48)) + list(range(96, 144)), 1: list(range(48, 96)) + list(range(144, 192))}
    allc = list(range(192))
    got = [ra.affinity(r, 8, nodes, node_cpus, allc) for r in range(8)]
    assert all(len(c) == 24 for c, _ in got) and [n for _, n in got] == nodes
    assert sorted(c for cs, _ in got for c in cs) == allc                    # disjoint, every core used once
    assert set(got[5][0]) <= set(node_cpus[1])
    even = [ra.affinity(r, 4, [None] * 4, {}, list(range(10))) for r in range(4)]
    assert [c for c, _ in even] == [[0, 1], [2, 3], [4, 5], [6, 7]] and all(n == -1 for _, n in even)


def test_single_call_operators_take_the_reference_modules_keys(synthetic_sd):
    """demfi_amd/ops.py: SepConvGRU / FGAC take the state_dict keys of the reference modules (DeMFInet.py:830-836, 369-380) and name the
    missing ones; no kernel is launched here."""
    from demfi_amd.ops import FGAC, SepConvGRU
    g = SepConvGRU(64, 64, device='cpu').load_state_dict(synthetic_sd, prefix='Booster_Module.GB.')
    assert sorted(g.sd) == sorted('conv%s%d.%s' % (a, i, p) for a in 'zrq' for i in (1, 2) for p in ('weight', 'bias'))
    assert tuple(g.sd['convz1.weight'].shape) == (64, 128, 1, 5) and tuple(g.sd['convq2.weight'].shape) == (64, 128, 5, 1)
    f = FGAC(device='cpu').load_state_dict(synthetic_sd, prefix='FAC_FB_Module.shared_FGAC.')
    assert 'conv_source_k.weight' in f.sd and tuple(f.sd['w_gen.weight'].shape) == (64, 128, 3, 3)
    with pytest.raises(KeyError, match='convr2.bias'):
        SepConvGRU(device='cpu').load_state_dict({k: v for k, v in synthetic_sd.items() if not k.endswith('convr2.bias')}, prefix='Booster_Module.GB.')
    with pytest.raises(RuntimeError):
        FGAC(device='cpu')(torch.zeros(1, 64, 8, 8), torch.zeros(1, 64, 8, 8), torch.zeros(1, 2, 8, 8))
    with pytest.raises(ValueError):
        SepConvGRU(dtype=torch.bfloat16)


def test_single_call_operator_plans_interpreted_on_the_cpu(synthetic_sd):
    """The launch plans demfi_amd/ops.py builds (fused z|r weights, GRU epilogue wiring, FGAC's conv chain) interpreted by the CPU plan
    interpreter agree with the oracle's sep_conv_gru / fgac: the host logic of the operators is right before a GPU sees it."""
    from demfi_amd.ops import FGAC, SepConvGRU
    torch.manual_seed(2)
    B, H, W = 2, 12, 20
    h, x = torch.tanh(torch.randn(B, 64, H, W)), torch.randn(B, 64, H, W)
    g = SepConvGRU(64, 64, dtype=torch.float32, device='cpu').load_state_dict(synthetic_sd, prefix='Booster_Module.GB.')
    pl, bufs, n = g._plan(B, H, W)
    bufs['h0'].copy_(h.permute(0, 2, 3, 1))
    bufs['x'].copy_(x.permute(0, 2, 3, 1))
    sim = PlanSim(pl)
    for i in range(n):
        sim.conv(pl._descs[i])
    assert (bufs['h2'].permute(0, 3, 1, 2) - O.sep_conv_gru(synthetic_sd, h, x)).abs().max() < 1e-5

    name = 'FAC_FB_Module.shared_FGAC'
    ref, src = torch.randn(B, 64, H, W), torch.randn(B, 64, H, W)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    flow = torch.stack([xs, ys])[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W) * 2.0
    f = FGAC(dtype=torch.float32, device='cpu').load_state_dict(synthetic_sd, prefix=name + '.')
    pl, bufs = f._plan(B, H, W)
    bufs['ref'].copy_(ref.permute(0, 2, 3, 1))
    bufs['source'].copy_(src.permute(0, 2, 3, 1))
    sim = PlanSim(pl)
    sim.conv(pl._descs[0])
    bufs['sampled'].copy_(O.fgac_sample(bufs['ref_k'].permute(0, 3, 1, 2), flow).permute(0, 2, 3, 1))     # the gather kernel's role
    for i in (1, 2, 3):
        sim.conv(pl._descs[i])
    want, w_want = O.fgac(synthetic_sd, name, ref, src, flow)
    w = bufs['w'].view(B, 1, H, W)
    assert (w - w_want).abs().max() < 1e-5
    out = w * src + (1 - w) * bufs['e_s'].permute(0, 3, 1, 2)                                             # the gate kernel's role
    assert (out - want).abs().max() < 1e-5


def test_per_t_contexts_stay_independent_in_the_arena(synthetic_sd):
    """The per-t contexts of a trunk set may run CONCURRENTLY on different streams (demfi_forward_t; the runner's DEMFI_TB=0 mode).
    With buffers sharing memory that only holds if everything context q touches lies in memory no other context touches (the
    slotted arena of ctx.cpp::plan_arena).  Interpreted on the CPU: the op lists of three contexts interleaved with a lag of
    11 ops between them leave exactly what each context leaves when it runs alone."""
    from demfi_amd.engine import SEG_HEAD, SEG_ITER, SEG_TRUNK
    H, W, N, NC = 32, 64, 2, 3
    eng = Engine(synthetic_sd, H, W, torch.float16, 'cpu', max_updates=N, n_ctx=NC)
    x = synthetic_window(H, W, 9)
    ts = [0.125, 0.5, 0.75]
    sim = PlanSim(eng)
    want = []
    for c, t in enumerate(ts):
        eng.use_ctx(c)
        sim.forward(x, t, N)
        want.append({k: eng._ctxs[0][c][k].clone() for k in ('finals', 'delta', 'occ', 'sharp1')})
        for k in ('finals', 'delta', 'occ', 'sharp1'):
            eng._ctxs[0][c][k].zero_()
    eng.use_ctx(0)
    eng.x.copy_(x[0])
    sim.run(eng.ops(SEG_TRUNK))
    seqs = []
    for c, t in enumerate(ts):
        eng._ctxs[0][c]['t_dev'].fill_(t)
        seqs.append(list(eng.ops(SEG_HEAD, c=c)) + [o for it in range(N) for o in eng.ops(SEG_ITER, it, c=c)])
    lag, n = 11, len(seqs[0])
    for i in range(n + lag * (NC - 1)):
        for c in range(NC):
            j = i - lag * c
            if 0 <= j < n:
                sim.run([seqs[c][j]])
    for c in range(NC):
        for k, v in want[c].items():
            assert torch.equal(eng._ctxs[0][c][k], v), (c, k)


def _plan_digest(env):
    """finals / delta of a CPU-interpreted fp16 forward + workspace sizes, in a fresh process (the layout switches are read once)."""
    import json
    import subprocess
    code = (
        "import hashlib, json, torch\n"
        "from demfi_amd import _lib as L\n"
        "from demfi_amd.engine import Engine\n"
        "from demfi_amd.weights import synthetic_state_dict, synthetic_window\n"
        "from tests.plan_sim import PlanSim\n"
        "eng = Engine(synthetic_state_dict(0), 32, 64, torch.float16, 'cpu', max_updates=2, n_ctx=2)\n"
        "sim = PlanSim(eng)\n"
        "sim.forward_tb(synthetic_window(32, 64, 4), [0.25, 0.75], 2)\n"
        "h = hashlib.sha256()\n"
        "for c in range(2):\n"
        "    for k in ('finals', 'delta', 'occ', 'sharp1'):\n"
        "        h.update(eng._ctxs[0][c][k].contiguous().numpy().tobytes())\n"
        "lib = L.load()\n"
        "print(json.dumps({'digest': h.hexdigest(), 'ws_small': int(eng.workspace.numel()),\n"
        "                  'ws_720p': int(lib.demfi_workspace_bytes(736, 1280, 3, L.F16, 3, 7)),\n"
        "                  'ws_1080p': int(lib.demfi_workspace_bytes(1088, 1920, 3, L.F16, 3, 5))}))\n")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_workspace_arena_shrinks_the_workspace_and_changes_no_result():
    """the per-t / trunk buffers are planned by liveness (ctx.cpp, plan_arena): the headline configuration needs
    <= 45 GB instead of 87.5 GB, config 5 (1080p x16) fits three trunk sets under the half-of-HBM rule, and the launch plan computes
    bit-identical results whether buffers share memory or not (batched per-t plan, fp16, two recursions, interpreted on the CPU --
    recycled memory holds the previous tenant's values there exactly as on the GPU)."""
    on, off = _plan_digest({'DEMFI_ARENA': '1'}), _plan_digest({'DEMFI_ARENA': '0'})
    assert on['digest'] == off['digest']
    assert on['ws_small'] < 0.7 * off['ws_small']
    assert on['ws_720p'] <= 45e9 < 80e9 < off['ws_720p']
    assert on['ws_1080p'] <= 144e9 * 0.5 < off['ws_1080p']
