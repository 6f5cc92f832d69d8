"""Clip-parallel multi-process path on CPU: gloo, world_size 2 (the 8-GPU run uses the same code over RCCL)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from demfi_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch as th
    from demfi_amd.engine import Engine
    from demfi_amd.spec import state_dict_shapes
    from demfi_amd.weights import synthetic_state_dict
    from demfi_amd import dist as DD
    th.set_num_threads(1)
    assert DD.init(world, rank, 0, backend='gloo')
    # rank 0 holds the weights, the other ranks start from zeros (as bench.py does) and receive ONE flat broadcast
    sd = synthetic_state_dict(0) if rank == 0 else {k: th.zeros(s) for k, s in state_dict_shapes().items()}
    eng = Engine(sd, 32, 32, th.float16, 'cpu', max_updates=1)
    before = int(eng.weight_blob.to(th.int64).sum())
    DD.broadcast_weights(eng, world)
    after = int(eng.weight_blob.to(th.int64).sum())
    lo, hi = DD.shard_windows(11, world, rank)
    tmax = DD.max_over_ranks(float(rank + 1), 'cpu')
    tot = DD.sum_over_ranks([hi - lo, 1.0], 'cpu')
    DD.barrier()
    q.put((rank, before, after, lo, hi, tmax, tot.tolist()))
    DD.finalize()


def test_two_process_gloo_broadcast_and_sharding():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, b0, a0, lo0, hi0, m0, t0), (_, b1, a1, lo1, hi1, m1, t1) = res
    assert b0 == a0 and b1 != a0 and a1 == a0              # rank 1 received rank 0's packed weights
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)           # contiguous blocks covering all windows once
    assert m0 == m1 == 2.0                                 # MAX over ranks (bench timing rule)
    assert t0 == t1 == [11.0, 2.0]


def test_shard_windows_partition():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 8):
            got = [D.shard_windows(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_noop():
    assert D.init(1, 0) is False
    assert D.max_over_ranks(3.5, 'cpu') == 3.5
    D.barrier()


def _clip_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    import torch as th
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from demfi_amd import DeMFInet, HyperParams, synthetic_state_dict, synthetic_window
    from demfi_amd.clip import window_list
    from demfi_amd.engine import Engine
    from demfi_amd import dist as DD
    from tests.plan_sim import PlanSim
    th.set_num_threads(1)
    assert DD.init(world, rank, 0, backend='gloo')
    # only rank 0 holds the checkpoint; ONE flat broadcast gives every rank the real state_dict
    model = DeMFInet(HyperParams(), dtype=th.float32)
    if rank == 0:
        model.load_state_dict(synthetic_state_dict(0))
    v0 = model._weights_version
    DD.broadcast_state_dict(model, world, device='cpu')
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    nonzero = sum(int(v.abs().sum() > 0) for k, v in sd.items() if k.endswith('.weight'))
    # an engine built AFTER the broadcast (what model.engine() does for a new frame size) packs real weights on rank 1 too
    eng = Engine(sd, 32, 32, th.float32, 'cpu', max_updates=1)
    blob_sum = int(eng.weight_blob.to(th.int64).sum())
    # a tiny clip, sharded: 6 frames -> 3 windows; every rank interprets the plan of ITS windows on CPU
    H = W = 32
    frames = [synthetic_window(H, W, 40 + i)[0, :, 0] for i in range(6)]         # [3,H,W] each
    wins = window_list(len(frames))
    lo, hi = DD.shard_windows(len(wins), world, rank)
    sim = PlanSim(eng)
    sums = th.zeros(len(wins), dtype=th.float64)
    for k in range(lo, hi):
        x = th.stack([frames[i] for i in wins[k]], 1)[None]                      # [1,3,4,H,W] in (B0,B1,B-1,B2) order
        sim.forward(x, 0.5, 1)
        sums[k] = eng.finals[0, 2].double().sum()
    tot = DD.sum_over_ranks(sums, 'cpu')                                         # every window filled by exactly one rank
    cnt = DD.sum_over_ranks([float(hi - lo)], 'cpu')
    q.put((rank, nonzero, blob_sum, model._weights_version - v0, tot.tolist(), cnt.tolist(), (lo, hi)))
    DD.finalize()


def test_two_process_gloo_state_dict_broadcast_and_sharded_clip():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_clip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1] == r1[1] == 130                           # all 130 weight tensors non-zero on BOTH ranks
    assert r0[2] == r1[2] != 0                             # identical packed blobs from engines built after the broadcast
    assert r0[3] == r1[3] == 1                             # the models know their weights changed
    assert r0[4] == r1[4] and all(v != 0 for v in r0[4])   # 3 windows, each computed once, same totals everywhere
    assert r0[5] == [3.0] and (r0[6], r1[6]) == ((0, 2), (2, 3))
    # single-process reference of the same clip
    import torch as th
    from demfi_amd import synthetic_state_dict, synthetic_window
    from demfi_amd.clip import window_list
    from demfi_amd.engine import Engine
    from tests.plan_sim import PlanSim
    eng = Engine(synthetic_state_dict(0), 32, 32, th.float32, 'cpu', max_updates=1)
    frames = [synthetic_window(32, 32, 40 + i)[0, :, 0] for i in range(6)]
    sim = PlanSim(eng)
    for k, w in enumerate(window_list(6)):
        sim.forward(th.stack([frames[i] for i in w], 1)[None], 0.5, 1)
        assert abs(float(eng.finals[0, 2].double().sum()) - r0[4][k]) < 2e-3       # MKLDNN sums differ with the thread count


def _table_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from demfi_amd import dist as DD
    from demfi_amd.clip import EvalTable
    assert DD.init(world, rank, 0, backend='gloo')
    scenes = ['s0', 's1', 's2']                                  # the common scene list; ranks own different windows / scenes
    t = EvalTable(4)
    if rank == 0:
        t.update('s0', 0, 30.0, 0.90)
        t.update('s0', 1, 31.0, 0.91)
    else:
        t.update('s0', 0, 32.0, 0.92)
        t.update('s2', 2, 40.0, 0.95)
        t.update('s2', 3, 25.0, 0.80)                            # a deblur column
    t.all_reduce(scenes, 'cpu')
    q.put((rank, t.summary()))
    DD.finalize()


def test_two_process_gloo_eval_table_all_reduce():
    """ranks hold different key sets; the reduced table must be the same, complete table on every rank."""
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_table_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s0, s1 = res[0][1], res[1][1]
    assert s0['samples'] == s1['samples'] == 4
    assert s0['per_index'][0] == s1['per_index'][0] == (31.0, 0.91)        # scene s0: mean(30, 32); only scene with column 0
    assert s0['per_index'][2] == (40.0, 0.95) and s0['deblur'] == s1['deblur'] == (25.0, 0.80)
    assert abs(s0['total'][0] - (30 + 31 + 32 + 40) / 4) < 1e-12
