"""GPU-box diagnostic: per-stage differences between the HIP engine buffers and the oracle's intermediates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch, numpy as np
from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
from oracle import demfi_oracle as O

H, W, seed, tval, N = [float(v) if '.' in v else int(v) for v in (sys.argv[1:6] or ['64', '96', '3', '0.875', '2'])]
dtype = torch.float16 if len(sys.argv) > 6 and sys.argv[6] == 'f16' else torch.float32
sd = synthetic_state_dict(0)
x = synthetic_window(H, W, seed)
t = torch.tensor([[tval]])
with torch.no_grad():
    ref, st = O.forward(sd, x, t, N, return_stages=True)
m = DeMFInet(HyperParams(), dtype=dtype); m.load_state_dict(sd); m = m.to('cuda:0').eval()
out = m(x.cuda(), t.cuda(), N)
e = m.engine(H, W, N)
nhwc = lambda z: z[0].permute(1, 2, 0)
def cmp(name, a, b):
    d = (a.float().cpu() - b).abs()
    print('%-9s max %.3e mean %.3e  frac>5e-4 %.5f  argmax %s' % (name, d.max(), d.mean(), (d > 5e-4).float().mean(),
          tuple(int(v) for v in np.unravel_index(int(d.argmax()), d.shape))))
cmp('F0', e.F01[0], nhwc(st['F0'])); cmp('flow_01', e.ffo[0:2], st['flow_01'][0]); cmp('flow_10', e.ffo[2:4], st['flow_10'][0])
cmp('occ', e.ffo[4:5], st['occ_logit'][0]); cmp('enc', e.enc, st['enc'].permute(0, 2, 3, 1))
cmp('aF0', e.aF[0], nhwc(st['aF0'])); cmp('aF1', e.aF[1], nhwc(st['aF1']))
cmp('flow_t0', e.ft[0:2], st['flow_t0'][0]); cmp('flow_t1', e.ft[2:4], st['flow_t1'][0]); cmp('Ft', e.Ft[0], nhwc(st['Ft']))
cmp('rF0', e.rF[0], nhwc(st['rF0'])); cmp('rFt', e.rF[2], nhwc(st['rFt']))
for i in range(3): cmp('d1_%d' % i, out[0][i][0], ref[0][i][0])
for i in range(3): cmp('fin_%d' % i, out[1][N - 1][i][0], ref[1][N - 1][i][0])
for i in range(N + 1): cmp('flows%d' % i, out[2][i][0], ref[2][i][0])
