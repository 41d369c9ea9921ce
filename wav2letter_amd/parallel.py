"""Data-parallel plumbing: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  Mirrors the reference's use of fl::distributed
(recipes/slimIPL/src/Train.cpp): initDistributed :188-196, allReduceParameters at start
:1078-1079, gradient reduction :1721-1735, batch-size all-reduce :1743-1747 -- but with ONE
collective per step over the flat gradient arena instead of dozens of ~20 MB buckets
(the batch-size scalar rides in the arena tail).
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None, device=None):
    """returns (rank, world). No-op for world == 1 (--enable_distributed=false)."""
    rank, local_rank, world = env_rank_world()
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_items, rank, world):
    """contiguous shard [lo, hi) of a global minibatch for this rank (createDataset(..., rank, world))"""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def sync_parameters(params):
    """allReduceParameters: make replicas identical (sum, then scale by 1/world)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return params
    dist.all_reduce(params)
    params.mul_(1.0 / dist.get_world_size())
    return params


class GradientArena:
    """flat gradient buffer [grads | 1 slot holding the local batch size]: one all-reduce sums both"""

    def __init__(self, n_floats, device):
        self.buf = torch.zeros(n_floats + 4, dtype=torch.float32, device=device)
        self.n = n_floats

    @property
    def grads(self):
        return self.buf[:self.n]

    def set_local_batch(self, b):
        self.buf[self.n] = float(b)

    def all_reduce(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.buf)
        return self.buf[self.n]  # total batch size (device scalar)


def bucket_offsets(param_table, total_floats, n_buckets):
    """Cut the flat gradient arena into n_buckets contiguous pieces of roughly equal size at PARAMETER
    boundaries.  param_table: [(name, numel, offset)] in arena order (Trainer.param_table()).  Returns
    ascending float offsets, offsets[0] == 0; bucket k = [offsets[k], offsets[k+1]), the last one runs to
    total_floats (criterion parameters ride in it)."""
    bounds = sorted({off for _, _, off in param_table} | {0})
    out = [0]
    for k in range(1, n_buckets):
        want = total_floats * k // n_buckets
        best = min(bounds, key=lambda b: abs(b - want))
        if best > out[-1]:
            out.append(best)
    return out


class OverlappedReducer:
    """Gradient all-reduce overlapped with the backward pass (the role of fl's CoalescingReducer,
    recipes/slimIPL/src/Train.cpp:195, :1721-1735, re-cut for one flat arena over RCCL/xGMI).

    Backward finalises the arena from its END towards its start; the C++ trainer records one HIP event per
    bucket on the compute stream.  After forward_backward has been ENQUEUED (it never blocks the host),
    reduce() walks the buckets last-to-first: a side stream waits for the bucket's event and the
    collective is issued on it, so the sum of the last layers' gradients crosses xGMI while the GPU is
    still computing the backward pass of the first layers.  Few large buckets (default 4, ~200 MB each for
    TDS-CTC): xGMI rings are per-link bound, large messages keep them at line rate."""

    def __init__(self, trainer, n_buckets=4):
        self.tr = trainer
        offs = bucket_offsets(trainer.param_table(), trainer.n_floats, n_buckets)
        trainer.set_grad_buckets(offs)
        self.offsets = offs + [trainer.n_floats]
        self.comm = torch.cuda.Stream(device=trainer.device)

    def reduce(self):
        """call right after trainer.forward_backward(); returns when every collective is enqueued and the
        current (compute) stream has been made to wait for them"""
        works = []
        g = self.tr.grads
        for k in reversed(range(len(self.offsets) - 1)):
            self.tr.wait_bucket(k, self.comm)
            with torch.cuda.stream(self.comm):
                works.append(dist.all_reduce(g[self.offsets[k]:self.offsets[k + 1]], async_op=True))
        for w in works:
            w.wait()
