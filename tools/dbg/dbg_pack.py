import sys, os
sys.path.insert(0, os.getcwd())
import torch
from demfi_amd import _lib as L
from demfi_amd.engine import Plan, _Dst
DEV='cuda:0'
def run(dsts, pack_ch, H, W, batch, use_pack=True):
    torch.manual_seed(23)
    pl = Plan(H, W, torch.float16, DEV)
    x = pl._fat(H, W, 32, batch); x.copy_(torch.randn(x.shape, device=DEV))
    rec = pl._fat(H, W, 16, batch)
    outs, D, c0, ress = [], [], 0, []
    for n, has_res in dsts:
        o = torch.zeros((batch * n, H, W), dtype=torch.float32, device=DEV)
        r = torch.randn((batch * n, H, W), dtype=torch.float32, device=DEV) if has_res else None
        outs.append(o)
        ress.append(r)
        sb = n * H * W if batch > 1 else 0
        D.append(_Dst(pl.tview(o, 0, sb=sb), range(c0, c0 + n), L.ACT_NONE, res=pl.tview(r, 0, sb=sb) if has_res else None))
        c0 += n
    wt = torch.randn(c0, 32, 3, 3) * (1.0 / (32 * 9) ** 0.5)
    bs = torch.randn(c0) * 0.1
    pl.conv([], 'thinpack', [pl.fsrc(x, 0)], D, H, W, batch=batch, weight=wt, bias=bs, pack=(pl.fview(rec), pack_ch) if use_pack else None)
    pl._upload()
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        rec.fill_(-3.0)
        pl.launch_conv(0, st)
    torch.cuda.synchronize()
    exp = torch.full((batch, H, W, 16), -3.0, dtype=torch.float16)
    for (n, _), o, chn in zip(dsts, outs, pack_ch):
        if chn < 0: continue
        n4 = (n + 3) // 4 * 4
        exp[..., chn:chn + n4] = 0.0
        exp[..., chn:chn + n] = o.view(batch, n, H, W).permute(0, 2, 3, 1).half().cpu()
    got = rec.cpu()
    F = torch.nn.functional
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.half().double(), bs.double(), padding=1)
    c1 = 0
    for (n, hr), o, r in zip(dsts, outs, ress):
        e = ref[:, c1:c1 + n]
        if hr: e = e + r.view(batch, n, H, W).double().cpu()
        g = o.view(batch, n, H, W).double().cpu()
        print('   dst n=%d: nan in x %d, res %d, out %d; max err vs torch %.3e' % (n, int(torch.isnan(x).sum()), int(torch.isnan(r).sum()) if hr else -1, int(torch.isnan(g).sum()), float((g - e).abs().nan_to_num(99.0).max())))
        c1 += n
    if not use_pack:
        return
    bad = (got != exp)
    print(dsts, pack_ch, H, W, batch, 'mismatches', int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print(' first', idx[:8].tolist(), 'channels', sorted(set(idx[:, 3].tolist())), 'images', sorted(set(idx[:,0].tolist())), 'rows', sorted(set(idx[:,1].tolist()))[:10], 'cols', sorted(set(idx[:,2].tolist()))[:10])
        i = idx[0].tolist(); print(' got', got[i[0], i[1], i[2]].tolist(), ' exp', exp[i[0], i[1], i[2]].tolist())
for dsts, pc in (([(5, True)], [0]), ([(4, True), (1, True)], [0, 4]), ([(3, False), (5, True)], [-1, 8])):
    for H, W, b in ((8, 32, 1), (37, 75, 2), (64, 96, 1), (37, 75, 1), (16, 64, 2)):
        run(dsts, pc, H, W, b)
        run(dsts, pc, H, W, b, use_pack=False)
