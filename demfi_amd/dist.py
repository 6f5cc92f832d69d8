"""Clip-parallel multi-GPU support: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The forward path has no cross-GPU dependence inside a window: windows are sharded in
contiguous blocks, the only collectives are ONE broadcast of the flat repacked weight blob at start-up and the
max/sum reductions of the bench bookkeeping.  On CPU (tests) the same code runs over gloo."""
import os

import torch
import torch.distributed as dist

_ACTIVE = False


def init(world, rank, local_rank=0, backend=None, force=False):
    """Initialise the process group when world > 1 (env:// rendezvous, MASTER_ADDR should be 127.0.0.1).  force: also for a world of
    one -- every collective below then really runs (tests/test_gpu_dist.py drives the RCCL code path on a single-GPU box that way)."""
    global _ACTIVE
    if world <= 1 and not force:
        return False
    if not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _ACTIVE = True
    return True


def active():
    return _ACTIVE and dist.is_initialized()


def shard_windows(n_windows, world, rank):
    """Static contiguous block of window indices for this rank: [lo, hi).  All t of a window stay on one GPU so the
    trunk cache is reused."""
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(engine, world, src=0):
    """One flat broadcast of the already-repacked weight/bias blob of an engine (~15 MB fp16 / 30 MB fp32).  Only the
    ENGINE's copy changes: a model whose parameters are not also synchronised would repack zeros the next time it builds
    an engine -- use ``broadcast_state_dict`` for models (bench.py, ClipRunner launches)."""
    if not active():
        return
    dist.broadcast(engine.weight_blob, src=src)


def broadcast_state_dict(model, world, src=0, device=None):
    """ONE flat broadcast (RCCL over xGMI on the GPU box, gloo in the CPU tests) of all 260 parameter tensors
    (7 408 284 fp32 = 29.6 MB) from rank ``src``; every rank then holds the real state_dict, so any engine it builds later
    (another frame size, more contexts) packs the right weights.  Bumps the model's weights version."""
    if not active():
        return
    params = [p for _, p in sorted(model.state_dict().items())]
    dev = device or params[0].device
    flat = torch.cat([p.detach().reshape(-1).to(dev, torch.float32) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p).to(p.device, p.dtype))
            off += n
    if hasattr(model, 'invalidate_weights'):
        model.invalidate_weights()


def barrier():
    if active():
        dist.barrier()


def max_over_ranks(value, device):
    if not active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(vec, device):
    """All-reduce (sum) of a small fp64 bookkeeping vector, e.g. [sum_psnr, frames, windows]."""
    t = torch.as_tensor(vec, dtype=torch.float64, device=device)
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def finalize():
    global _ACTIVE
    if active():
        dist.destroy_process_group()
    _ACTIVE = False
