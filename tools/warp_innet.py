# in-network timing of warp_blend_fat (real flows of the network), per variant
import os, sys, torch
sys.path.insert(0, os.getcwd())
from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
m = DeMFInet(HyperParams(), dtype=torch.float16); m.load_state_dict(synthetic_state_dict(0)); m = m.to('cuda:0').eval()
x = synthetic_window(736, 1280, 1).to('cuda:0')
m(x, torch.tensor([[0.5]], device='cuda:0'), 3)
eng = m.engine(736, 1280, 3)
prof = eng.profile(3, reps=10, isolated=bool(int(os.environ.get('ISOLATED', '0'))))
for p in prof:
    if p[1] in ('warp_fat',): print(os.environ.get('DEMFI_WARP_VAR','0'), p[1], '%.4f ms' % p[3], '%.1f GB/s' % (404*736*1280/p[3]/1e6))
