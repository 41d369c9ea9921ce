#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native wav2letter acoustic-training hot path.

Metric (BASELINE.json): utterances/sec (whole job) on the TDS-CTC LibriSpeech training
step -- sota/2019 am_tds_ctc.arch, 80-mel x T=1500, 9998 classes (10k word pieces + blank),
batch 32 per GPU, fp32 -- plus the ASG criterion time (ms/step) at the conv_glu
LibriSpeech criterion shape (B=64, T=2000, N=30).

A "step" = SpecAugment + network forward + CTC forward/backward + network backward +
(N>1: RCCL all-reduce of the flat gradient arena in a few large buckets, issued on a side
stream under the backward pass) + gradient clipping + SGD with
momentum, exactly the reference's hot loop (recipes/slimIPL/src/Train.cpp:1454-1804).
Inputs are synthetic, generated on the device before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBPS = 8000.0         # same table (6.29 TB/s measured-achievable)


def pmc_traffic(key, fallback=None):
    """HBM-side bytes per launch of a kernel family from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, written by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes over this same bench command; gfx950 corrections applied there).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        if key not in d and fallback:
            key = fallback
        return round(d[key]["hbm_bytes_per_launch"]), {
            "fetch_bytes": round(d[key]["fetch_bytes_per_launch"]), "write_bytes": round(d[key]["write_bytes_per_launch"] or 0),
            "source": "profiles/" + os.path.basename(files[-1]),
            "note": "per launch; FETCH_SIZE x2 (gfx950 half-count) + WRITE_SIZE = L2->fabric requests, Infinity-Cache hits included"}
    except (KeyError, ValueError, OSError):
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--buckets", type=int, default=4, help="gradient all-reduce buckets (N > 1)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL path even with one rank (smoke test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-asg", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the ASG N=9998 stress leg")
    ap.add_argument("--stress-frames", type=int, default=1500)
    ap.add_argument("--no-c4", action="store_true", help="skip the conv_glu LibriSpeech ASG step (BASELINE config 4) leg")
    return ap.parse_args()


def make_batch(B, T, nfeat, nlabel, Lmax, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, nfeat, T, generator=g, dtype=torch.float32)  # reference input (T,NFEAT,1,B)
    tgt = torch.full((B, Lmax), -1, dtype=torch.int32)
    for b in range(B):
        l = int(torch.randint(20, Lmax + 1, (1,), generator=g))
        tgt[b, :l] = torch.randint(0, nlabel - 1, (l,), generator=g, dtype=torch.int32)
    return x.to(device), tgt.to(device)


def _timeit(fn, n=10, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def asg_criterion_ms(device):
    """ASG (FCC + FAC) forward and forward+backward at the conv_glu LibriSpeech criterion shape
    (BASELINE config C4: B=64/GPU, T=2000, N=30 letters, --transdiag=4, target/sqrt scaling)"""
    from wav2letter_amd import ASGLoss, CriterionScaleMode
    B, T, N, L = 64, 2000, 30, 300
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, T, N, generator=g).to(device).requires_grad_(True)
    tgt = torch.full((B, L), -1, dtype=torch.int32)
    for b in range(B):
        l = int(torch.randint(60, L + 1, (1,), generator=g))
        y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
        for i in range(1, l):  # replabel convention: no identical neighbours
            if y[i] == y[i - 1]:
                y[i] = (y[i] + 1) % 28
        tgt[b, :l] = y
    tgt = tgt.to(device)
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).to(device)
    fwd = _timeit(lambda: crit(x, tgt))
    fb = _timeit(lambda: crit(x, tgt).sum().backward())
    fcc_f = _timeit(lambda: crit.fcc(x, tgt))
    fac_f = _timeit(lambda: crit.fac(x, tgt))
    vit = _timeit(lambda: crit.viterbiPath(x.detach()))
    return {"shape": f"B={B},T={T},N={N},L<={L}", "fwd_ms": round(fwd, 4), "fwd_bwd_ms": round(fb, 4),
            "fcc_fwd_ms": round(fcc_f, 4), "fac_fwd_ms": round(fac_f, 4), "viterbi_ms": round(vit, 4),
            "algorithmic_bytes": 16 * B * T * N, "achieved_GBps": round(16 * B * T * N / (fb * 1e-3) / 1e9, 2),
            "us_per_frame_fwd": round(fwd * 1e3 / T, 3),
            "note": "T dependent steps per utterance: latency-bound at N=30 (SURVEY 8d), HBM time would be ~10 us; "
                    "the HBM roofline is quoted on the N=9998 stress shape (asg_stress)"}


def conv_glu_asg_step(device, L, steps=2):
    """BASELINE config 4 on one GPU: conv_glu LibriSpeech (17 WN-conv + GLU layers, 208.9 M parameters), ASG criterion,
    N = 30, T = 2000 frames of 40 filterbanks, batch 64: full training step (forward, ASG, backward, clip + SGD).  The
    convolutions run as one LDS-DMA GEMM each on overlapping rows of the frame-major activations (csrc/conv.hip)."""
    import ctypes as C
    from wav2letter_amd import CriterionScaleMode, recipes
    from wav2letter_amd.trainer import Trainer
    B, T, nfeat, nlabel, Lmax = 64, 2000, 40, 30, 300
    fl = recipes.CONV_GLU_FLAGS
    tr = Trainer(recipes.conv_glu_librispeech_arch(), nfeat, nlabel, "asg", CriterionScaleMode.TARGET_SZ_SQRT,
                 transdiag=fl["transdiag"], device=device)
    tr.init_params(seed=1)
    tr.plan(B, T, Lmax)
    tr.to_device()
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, nfeat, T, generator=g).to(device)
    tgt = torch.full((B, Lmax), -1, dtype=torch.int32)
    for b in range(B):
        l = int(torch.randint(60, Lmax + 1, (1,), generator=g))
        y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
        for i in range(1, l):
            if y[i] == y[i - 1]:
                y[i] = (y[i] + 1) % 28
        tgt[b, :l] = y
    tgt = tgt.to(device)

    def step():
        loss = tr.forward_backward(x, tgt)
        tr.update(lr=fl["lr"], lrcrit=fl["lrcrit"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
        return loss
    step()
    torch.cuda.synchronize()
    L.w2l_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
    L.w2l_profile_report_kind(0, C.byref(n_), C.byref(ms_), C.byref(w_))
    L.w2l_profile_enable(0)
    tf = w_.value / (ms_.value * 1e-3) / 1e12 if ms_.value > 0 else 0.0
    return {"config": "conv_glu LibriSpeech ASG (recipes/conv_glu/librispeech/network.arch): B=64/GPU, T=2000, 40 fbank, N=30, fp32",
            "ms_per_step": round(dt * 1e3, 1), "utterances_per_sec": round(B / dt, 2), "finite": bool(torch.isfinite(loss).all().item()),
            "roofline": {"bound": "mfma", "kernel": "gemm128g_kernel / gemm160_kernel on overlapping-row convolution operands",
                         "achieved": round(tf, 1), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         "launches_per_step": n_.value // steps, "gemm_ms_per_step": round(ms_.value / steps, 1),
                         "algorithmic_tflop_per_step": round(w_.value / steps / 1e12, 2)}}


def asg_stress(device, L, T):
    """The north-star stress shape of the ASG alpha/beta recursion: B=32, T=1500, N=9998 word pieces.
    The 400 MB transition matrix exceeds the 256 MiB Infinity Cache and is re-streamed at every one of the
    T dependent steps: algorithmic bytes per step = 4 N^2 + 8 B N.  The dominant kernel
    (fcc_big_gemm, one launch per step) is timed with HIP events on its own stream inside the library."""
    import ctypes as C
    from wav2letter_amd import CriterionScaleMode, FullConnectionCriterion
    B, N = 32, 9998
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(B, T, N, generator=g).to(device).requires_grad_(True)
    tgt = torch.zeros(B, 8, dtype=torch.int32, device=device)
    crit = FullConnectionCriterion(N, CriterionScaleMode.TARGET_SZ_SQRT).to(device)
    crit.transitions.data = (torch.randn(N, N, generator=g) * 0.1 + 4.0 * torch.eye(N)).to(device)
    loss = crit(x, tgt)          # warm-up (allocates the 6 GB workspace)
    loss.sum().backward()
    del loss
    x.grad = None                # the timed backward reuses the warm-up's gradient buffers instead of allocating 2.3 GB
    crit.transitions.grad = None
    torch.cuda.synchronize()
    L.w2l_profile_enable(1)
    t0 = time.perf_counter()
    loss = crit(x, tgt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.sum().backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    nl, ms, by = C.c_int(0), C.c_double(0), C.c_double(0)
    L.w2l_profile_report_kind(3, C.byref(nl), C.byref(ms), C.byref(by))
    L.w2l_profile_enable(0)
    achieved = by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0
    step_bytes = 4.0 * N * N + 8.0 * B * N
    fwd_ms, bwd_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
    return {"shape": f"B={B},T={T},N={N}", "fwd_ms": round(fwd_ms, 2), "bwd_ms": round(bwd_ms, 2),
            "fwd_us_per_step": round(fwd_ms * 1e3 / T, 2),
            "whole_forward_GBps": round(step_bytes * (T - 1) / (fwd_ms * 1e-3) / 1e9, 1),
            "finite": bool(torch.isfinite(loss).all().item()),
            "roofline": {"bound": "hbm", "kernel": "fcc_big_gemm (packed-transition stream, fp32 MFMA 32x32x2)",
                         "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM_GBPS, 4), "traffic": pmc_traffic("fcc_big_gemm")[0],
                         "traffic_detail": pmc_traffic("fcc_big_gemm")[1],
                         "launches": nl.value, "avg_launch_us": round(ms.value * 1e3 / max(1, nl.value), 2),
                         "algorithmic_bytes_per_launch": step_bytes}}


def cpu_baseline(nfeat, nlabel, T):
    """the oracle ("port") timed on this box's host cores on a bounded sample: ONE utterance,
    network forward+backward + CTC, same arch and shapes (B=1)."""
    from oracle import pyoracle as O
    from oracle import refnet
    from wav2letter_amd import recipes
    cores = O.num_threads()
    Ts = T if cores >= 32 else max(200, T // 8)
    arch = recipes.tds_ctc_arch()
    arch = "\n".join(l for l in arch.splitlines()
                     if not l.startswith("SAUG"))  # eval-style pass, dropout-free lines rewritten below
    arch = "\n".join((" ".join(f[:4] + ["0.0"] + f[5:]) if f and f[0] == "TDS" else " ".join(f))
                     for f in (l.split() for l in arch.splitlines())) + "\n"
    net = refnet.RefNet(arch, nfeat, nlabel)
    rng = np.random.default_rng(0)
    params = net.random_params(rng)
    x = rng.normal(size=(1, 1, nfeat, Ts)).astype(np.float32)
    tgt = rng.integers(0, nlabel - 1, size=(1, 20)).astype(np.int32)
    t0 = time.perf_counter()
    em = net.forward(x, params)
    ctc = O.CTC(em, tgt, scale_mode=4)
    ctc.forward()
    d_em = ctc.backward().astype(np.float32)
    net.backward(d_em, len(params))
    dt = time.perf_counter() - t0
    return {"value": round((Ts / T) / dt, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"1 utterance x {Ts} frames (scaled to T={T}), forward+backward+CTC, no optimizer, "
                      f"oracle C loops with OpenMP; {dt:.1f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        a.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from wav2letter_amd import CriterionScaleMode, _lib, recipes
    from wav2letter_amd.trainer import Trainer

    nfeat, nlabel, Lmax = 80, 9998, 80
    B, T = a.batch, a.frames
    fl = recipes.TDS_CTC_FLAGS
    tr = Trainer(recipes.tds_ctc_arch(), nfeat, nlabel, "ctc", CriterionScaleMode.TARGET_SZ_SQRT, device=device)
    tr.init_params(seed=1)  # identical replicas on every rank (== allReduceParameters at start)
    Tout = tr.plan(B, T, Lmax)
    tr.to_device()
    x, tgt = make_batch(B, T, nfeat, nlabel, Lmax, 2026 + rank, device)
    total_batch = B * world

    reducer = None
    if dist is not None:
        from wav2letter_amd.parallel import OverlappedReducer
        reducer = OverlappedReducer(tr, n_buckets=a.buckets)

    def step():
        loss = tr.forward_backward(x, tgt)
        if reducer is not None:
            # the flat gradient arena (814 MB fp32) in a few large buckets, last layers first, on a side
            # stream gated by per-bucket events: the sum crosses xGMI under the rest of the backward pass
            reducer.reduce()
        tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"],
                  total_batch=total_batch)
        return loss

    for _ in range(a.warmup):
        step()
    L = _lib.lib()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        L.w2l_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    last_loss = float(loss.float().mean().item())
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    def kind(k):
        n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
        L.w2l_profile_report_kind(k, C.byref(n_), C.byref(ms_), C.byref(w_))
        return n_.value, ms_.value, w_.value
    nl, ms, flops = kind(0)          # the dominant kernel: 128x128 fp32 MFMA GEMM (+ its stream-K fix-up)
    nc, msc, flc = kind(2)           # TDS slab convolutions
    ns, mss, fls = kind(1)           # generic skinny implicit GEMM (strided backward-data only)
    L.w2l_profile_enable(0)
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    out = {
        "metric": "utterances/sec", "value": round(total_batch * a.steps / dt, 3), "unit": "utterances/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "seq2seq_tds LibriSpeech TDS-CTC (sota/2019 am_tds_ctc.arch): 80-mel x T=%d, "
                               "9998 classes, batch %d/GPU, fp32, SGD+momentum, CTC" % (T, B),
                   "global_batch": total_batch, "frames": T, "emission_frames": Tout, "parallelism": f"dp{world}",
                   "params": int(tr.n_net), "final_loss": round(last_loss, 4)},
        "roofline": {"bound": "mfma", "kernel": "gemm128g_kernel / gemm160_kernel (fp32 v_mfma_f32_32x32x2_f32; 128x128x32 tiles, or 128x160 / 160x128 where 128 leaves a ragged tile column; persistent, buffer LDS-DMA staging, stream-K tail reduced in-kernel; includes the few gemm128_kernel launches on unaligned shapes)",
                     "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic("gemm_lds_dma", "gemm128g")[0],
                     "traffic_detail": pmc_traffic("gemm_lds_dma", "gemm128g")[1],
                     "launches_per_step": nl // max(1, a.steps),
                     "avg_launch_us": round(ms * 1e3 / max(1, nl), 1),
                     "algorithmic_gflop_per_launch": round(flops / max(1, nl) / 1e9, 2),
                     "gemm_ms_per_step": round(ms / max(1, a.steps), 3),
                     "algorithmic_tflop_per_step": round(flops / max(1, a.steps) / 1e12, 3),
                     "tds_conv": {"launches_per_step": nc // max(1, a.steps), "ms_per_step": round(msc / max(1, a.steps), 3),
                                  "achieved_TFLOPs": round(flc / (msc * 1e-3) / 1e12, 2) if msc > 0 else None,
                                  "note": "N = C_out (10/14/18) padded to 16/16/32 MFMA columns: ceiling 62.5/87.5/56 % of peak"},
                     "skinny_gemm": {"launches_per_step": ns // max(1, a.steps), "ms_per_step": round(mss / max(1, a.steps), 3)}},
    }
    def leg(key, fn):
        # the additional legs never take the headline line down with them: a failure is reported in place
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    if not a.no_asg:
        leg("asg_loss_ms_per_step", lambda: asg_criterion_ms(device))
    if world == 1 and not a.no_stress:
        del tr, x, tgt
        torch.cuda.empty_cache()
        leg("asg_stress", lambda: asg_stress(device, L, a.stress_frames))
    if world == 1 and not a.no_c4:
        leg("conv_glu_asg_step", lambda: conv_glu_asg_step(device, L))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(nfeat, nlabel, T)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
