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
