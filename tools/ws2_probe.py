import sys, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demfi_amd import _lib as L
from demfi_amd.engine import Plan, _Dst
DEV='cuda:0'
H,W,batch=736,1280,7
pl = Plan(H, W, torch.float16, DEV)
h = pl._fat(H, W, 64, batch); r16 = pl._fat(H, W, 16, batch); a8 = pl._fat(H, W, 8, batch)
for b in (h, r16, a8): b.copy_(torch.randn(b.shape, device=DEV))
gw = pl._fat(H, W, 64, 1); gw.copy_(torch.randn(gw.shape, device=DEV))
out = pl._fat(H, W, 64, batch)
m16 = [64, 65, 66, 67, 68, 69, -1, -1, -1, 71, 72, 73, 74, -1, 70, -1]
wt = torch.randn(64, 83, 3, 3) * 0.03; bs = torch.randn(64) * 0.1
res = pl.fview(gw); res.sb = 0
pl.conv([], 'Dec_first_2#t', [pl.fsrc(h, 0), pl.fsrc_map(r16, m16, b=None), pl.fsrc_map(a8, list(range(75, 83)), b=None)],
        [_Dst(pl.fview(out), range(64), L.ACT_RELU, res=res)], H, W, batch=batch, weight=wt, bias=bs)
pl._upload()
st = torch.cuda.current_stream().cuda_stream
for i in range(3):
    pl.launch_conv(0, st)
torch.cuda.synchronize()
