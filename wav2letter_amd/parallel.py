"""Data-parallel plumbing: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  Mirrors the reference's use of fl::distributed
(recipes/slimIPL/src/Train.cpp): initDistributed :188-196, allReduceParameters at start
:1078-1079, gradient reduction :1721-1735, batch-size all-reduce :1743-1747 -- but with ONE
collective per step over the flat gradient arena instead of dozens of ~20 MB buckets
(the batch-size scalar rides in the arena tail).
"""
import contextlib
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


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks, script, argv, need_gpus=True):
    """Start `script argv` as n_ranks processes on this node, one per GPU, and return the exit code -- what the
    reference's users do with `mpirun -n 8 Train ...` (recipes/slimIPL/src/Train.cpp:188-196 initDistributed reads the
    MPI world).  Called by an entry point that was asked for N > 1 devices without a WORLD_SIZE in its environment, so
    that `python bench.py --gpus 8` IS an 8-rank RCCL job.  Rendezvous on 127.0.0.1 (the container hostname may not
    resolve).  Fails loudly when fewer than n_ranks devices are visible."""
    import subprocess
    import sys
    if n_ranks < 2:
        raise ValueError("self_launch: n_ranks must be >= 2")
    if need_gpus:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_ranks:
            raise RuntimeError(f"--gpus {n_ranks} needs {n_ranks} visible GPUs, this node shows {have} "
                               "(one process per GPU; set --gpus to the number of devices or fix HIP_VISIBLE_DEVICES)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env)


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

    def __init__(self, trainer, n_buckets=4, bf16=False):
        self.tr = trainer
        offs = bucket_offsets(trainer.param_table(), trainer.n_floats, n_buckets)
        # the last bucket runs over the 4-float tail too: tail[0] = local batch size, summed with the gradients
        # (the reference all-reduces the batch size on its own, recipes/slimIPL/src/Train.cpp:1743-1747)
        self.offsets = offs + [trainer.grads_full.numel()]
        # A gradient arena on the host (the world_size-2 gloo tests of the bucket walk on the REAL arena layout,
        # tests/test_distributed_cpu.py) has no streams and no bucket events: the same walk, issued synchronously.
        self.on_gpu = trainer.grads_full.is_cuda
        if self.on_gpu:
            trainer.set_grad_buckets(offs)
            self.comm = torch.cuda.Stream(device=trainer.device)
        else:
            trainer.bucket_offsets = list(offs)
            self.comm = None
        # bf16 buckets (the mixed-precision mode: fl's AMP all-reduces half-precision gradients, fp32 master weights stay local;
        # recipes/slimIPL/src/Train.cpp:1681-1760): a bucket's gradients are rounded to bf16 on the side stream, summed in
        # bf16 -- half the bytes over xGMI -- and written back as fp32; the 4-float tail (the batch size must stay exact) is
        # reduced in fp32 by a collective of its own.
        self.bf16 = bool(bf16)
        self._stage = {}
        # self-diagnosis of a multi-GPU run (bench.py --gpus N): with timing on, reduce() brackets every bucket's collective with
        # events on the side stream and marks where the compute stream starts to wait for the last of them
        self.timing = False
        self._ev = None

    def enable_timing(self, on=True):
        self.timing = bool(on) and self.on_gpu
        self._ev = None

    def timings(self):
        """after a synchronize: what the LAST timed reduce() cost -- per-bucket collective durations on the side stream in issue
        order (last layers first; a duration includes the wait for the slowest peer) and the exposed communication time = how
        long the compute stream sat at the join behind the backward pass.  None without timing / on the host path."""
        if not self._ev:
            return None
        pairs, reach, done = self._ev
        return {"bucket_allreduce_us": [round(a.elapsed_time(b) * 1e3, 1) for a, b in pairs],
                "bucket_bytes": self.bucket_bytes(),
                "comm_exposed_ms": round(max(0.0, reach.elapsed_time(done)), 4)}

    def _bf16_stage(self, k, n):
        b = self._stage.get(k)
        if b is None or b.numel() != n:
            b = torch.empty(n, dtype=torch.bfloat16, device=self.tr.grads_full.device)
            self._stage[k] = b
        return b

    def bucket_bytes(self):
        """bytes each collective of a step moves, in issue order (last bucket first; bf16 mode: the fp32 tail collective
        follows the last bucket's)"""
        nf = self.tr.n_floats
        out = []
        for k in reversed(range(len(self.offsets) - 1)):
            lo, hi = self.offsets[k], self.offsets[k + 1]
            if not self.bf16:
                out.append(4 * (hi - lo))
            else:
                out.append(2 * (min(hi, nf) - lo))
                if hi > nf:
                    out.append(4 * (hi - nf))
        return out

    def reduce(self):
        """call right after trainer.forward_backward(); returns when every collective is enqueued and the
        current (compute) stream has been made to wait for them"""
        works = []
        pairs = []
        g = self.tr.grads_full
        nf = self.tr.n_floats
        for k in reversed(range(len(self.offsets) - 1)):
            lo, hi = self.offsets[k], self.offsets[k + 1]
            if self.on_gpu:
                self.tr.wait_bucket(k, self.comm)
            ctx = torch.cuda.stream(self.comm) if self.on_gpu else contextlib.nullcontext()
            with ctx:
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    pairs.append((e0, e1))
                    e0.record(self.comm)
                if not self.bf16:
                    w = dist.all_reduce(g[lo:hi], async_op=True)
                    if self.timing:
                        # (the collective runs on the backend's own stream: the side stream joins it so that the closing event
                        #  lies behind it; the compute stream then waits for the side stream instead of for the work objects)
                        w.wait()
                        e1.record(self.comm)
                    else:
                        works.append(w)
                    continue
                hg = min(hi, nf)
                stage = self._bf16_stage(k, hg - lo)
                stage.copy_(g[lo:hg])                       # round to nearest even
                w = dist.all_reduce(stage, async_op=True)
                w.wait()                                    # stream-ordered on the GPU, blocking on the host
                g[lo:hg].copy_(stage)
                if hi > nf:
                    w = dist.all_reduce(g[nf:hi], async_op=True)
                    if self.timing:
                        w.wait()
                    else:
                        works.append(w)
                if self.timing:
                    e1.record(self.comm)
        if self.timing:
            reach, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reach.record(torch.cuda.current_stream(self.tr.device))   # the backward pass has been enqueued up to here
            done.record(self.comm)                                    # behind the last collective
            self._ev = (pairs, reach, done)
        for w in works:
            w.wait()
        if self.on_gpu and (self.bf16 or self.timing):
            torch.cuda.current_stream(self.tr.device).wait_stream(self.comm)
