import torch, sys
sys.path.insert(0, '/root/repo')
from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict
from demfi_amd.runner import WindowRunner
m = DeMFInet(HyperParams(gpu=0), dtype=torch.float16); m.load_state_dict(synthetic_state_dict(0)); m = m.to('cuda:0').eval()
r = WindowRunner(m, 720, 1280, 3, 8)
e = r.engine
print('contexts: trunk %d x per-t %d; activation bytes %.2f GB; torch allocated %.2f GB' % (e.n_trunk, e.n_ctx, e.activation_bytes() / 1e9, torch.cuda.memory_allocated() / 1e9))
