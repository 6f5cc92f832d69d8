"""The RCCL code path on real hardware (VERDICT r5 missing #2 / next #7): a single-GPU box cannot measure scaling, but it can run every
collective the clip-parallel launch issues -- backend "nccl" (= RCCL on ROCm), device tensors -- in a process group of ONE rank, so that
the first RCCL call this code ever makes is not the one on the driver's 8-GPU node."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
import torch.distributed as td
from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
from demfi_amd import dist as D
torch.cuda.set_device(0)
assert D.init(1, 0, 0, backend='nccl', force=True) and D.active()
assert td.get_backend() == 'nccl' and td.get_world_size() == 1
m = DeMFInet(HyperParams(), dtype=torch.float16)
sd = synthetic_state_dict(0)
m.load_state_dict(sd)
m = m.to('cuda:0').eval()
v0 = m._weights_version
D.broadcast_state_dict(m, 1, src=0, device='cuda:0')          # ONE flat 29.6 MB RCCL broadcast of the 260 tensors, staged on the GPU
same = all(torch.equal(p.cpu(), sd[k]) for k, p in m.state_dict().items())
x = synthetic_window(32, 64, 1).to('cuda:0')
out = m(x, torch.tensor([[0.5]], device='cuda:0'), 1)          # an engine built AFTER the broadcast packs the broadcast weights
eng = m.engine(32, 64, 1)
blob0 = int(eng.weight_blob.to(torch.int64).sum())
D.broadcast_weights(eng, 1)                                    # the packed blob (device memory) through RCCL
blob1 = int(eng.weight_blob.to(torch.int64).sum())
tmax = D.max_over_ranks(3.25, 'cuda:0')
tot = D.sum_over_ranks([5.0, 7.0, 1.0], 'cuda:0')
g = [torch.zeros(4, device='cuda:0')]
td.all_gather(g, torch.arange(4, dtype=torch.float32, device='cuda:0'))
D.barrier()
print('RESULT ' + json.dumps(dict(same=same, bumped=m._weights_version > v0, finite=bool(torch.isfinite(out[1][0][2]).all()), blob_same=blob0 == blob1 and blob0 != 0,
                      tmax=tmax, tot=tot.tolist(), tot_dev=str(tot.device), gathered=g[0].tolist(), shard=D.shard_windows(11, 1, 0))))
D.finalize()
assert not D.active()
"""


def test_rccl_collectives_in_a_one_rank_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-c', WORKER % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
    assert lines, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[-1][7:])
    assert d['same'] and d['bumped'] and d['finite'] and d['blob_same']
    assert d['tmax'] == 3.25 and d['tot'] == [5.0, 7.0, 1.0] and d['tot_dev'].startswith('cuda')
    assert d['gathered'] == [0.0, 1.0, 2.0, 3.0] and d['shard'] == [0, 11]
