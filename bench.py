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
  python bench.py --gpus N ...            (no WORLD_SIZE in the environment: starts N ranks itself, one per GPU)
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

PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline includes 2:1 sparsity)
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBPS = 8000.0         # same table (6.29 TB/s measured-achievable)


def pmc_traffic(key, fallback=None):
    """HBM-side bytes per launch of a kernel family from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, written by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes over this same bench command; gfx950 corrections applied there).  None if absent."""
    import glob
    import re

    def order(p):   # r<round>_run<n>_...: the newest counter summary of the newest round (NOT alphabetical: run17 > run9)
        m = re.match(r"r(\d+)_run(\d+)_", os.path.basename(p))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(set(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", "*_pmc_headline*.json"))), key=order)
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
    ap.add_argument("--no-c3", action="store_true", help="skip the streaming TDS fp32 / bf16 step (BASELINE config 3) leg")
    ap.add_argument("--no-input-pipeline", action="store_true", help="skip the prefetching-loader leg (SURVEY 8 f3)")
    ap.add_argument("--no-c5", action="store_true", help="skip the Transformer-CTC fp32 / bf16 step (BASELINE config 5) leg")
    ap.add_argument("--no-oracle-checks", action="store_true", help="skip the oracle comparison of the stress / config-4 losses")
    ap.add_argument("--cpu-baseline-batch", type=int, default=8)
    ap.add_argument("--dist-selftest", action="store_true",
                    help="launcher + process group + arena all-reduce only (gloo on a CPU-only host, RCCL on GPUs)")
    return ap.parse_args()


_T0 = time.perf_counter()


def note(msg):
    """progress on stderr (the JSON line is the only thing on stdout): which leg is running, since when"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


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


def ctc_criterion_ms(device, L):
    """CTC forward + backward through the C ABI at the two shapes SURVEY 8(d) names: the model-shaped criterion of the
    headline step (B=32, T'=188, N=9998) and the north-star stress (T=1500): 12 B T N algorithmic bytes (emissions read
    for the row normalisers, read again and the gradient written); preallocated buffers, HIP events around n calls"""
    from wav2letter_amd import criterion as Cr
    out = {}
    for name, (B, T, N, Lt, n) in {"model_shaped": (32, 188, 9998, 80, 20), "stress_T1500": (32, 1500, 9998, 80, 5)}.items():
        g = torch.Generator(device="cpu").manual_seed(11)
        x = torch.randn(B, T, N, generator=g).to(device)
        tgt = torch.full((B, Lt), -1, dtype=torch.int32)
        for b in range(B):
            l = int(torch.randint(20, Lt + 1, (1,), generator=g))
            tgt[b, :l] = torch.randint(0, N - 1, (l,), generator=g, dtype=torch.int32)
        tgt = tgt.to(device)
        ts = Cr.batch_target_size(tgt, T, ctc=True)
        ws = torch.empty(L.w2l_ctc_workspace_size(B, T, N, Lt), dtype=torch.uint8, device=device)
        loss = torch.empty(B, device=device)
        grad = torch.ones(B, device=device)
        dx = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream

        def fwd():
            _lib_check(L.w2l_ctc_forward(B, T, N, Lt, 4, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), loss.data_ptr(), ws.data_ptr(), st))

        def bwd():
            _lib_check(L.w2l_ctc_backward(B, T, N, Lt, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), grad.data_ptr(), dx.data_ptr(), ws.data_ptr(), st))
        f_ms = _timeit(fwd, n=n)
        fb_ms = _timeit(lambda: (fwd(), bwd()), n=n)
        by = 12.0 * B * T * N
        out[name] = {"shape": f"B={B},T={T},N={N},L<={Lt}", "fwd_ms": round(f_ms, 4), "fwd_bwd_ms": round(fb_ms, 4),
                     "algorithmic_bytes": by, "achieved_GBps": round(by / (fb_ms * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(by / (fb_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                     "finite": bool(torch.isfinite(loss).all().item())}
        del x, dx, ws
        torch.cuda.empty_cache()
    out["note"] = ("ctc_rows_lse (one streaming pass) + ctc_scan (alpha and beta side by side, one wave each per utterance) + "
                   "ctc_rows_grad (one streaming pass + occupancies); the scans are T dependent steps and do not stream")
    return out


def _lib_check(st):
    if st != 0:
        raise RuntimeError(f"C ABI call failed with status {st}")


def conv_glu_asg_step(device, L, steps=2, oracle_checks=True):
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
    check = None
    if oracle_checks:
        # outside the timed region: the criterion of the step on the network's own emissions (eval forward) against the
        # fp64 oracle at the full shape -- loss [64] element by element (reference call site Train.cpp:408-410, :1675)
        from oracle import pyoracle as O
        from wav2letter_amd import ASGLoss
        em = tr.forward(x, train=False).clone()
        A = tr.params[tr.n_net:tr.n_net + nlabel * nlabel].view(nlabel, nlabel).clone()
        crit = ASGLoss(nlabel, CriterionScaleMode.TARGET_SZ_SQRT, fl["transdiag"]).to(device)
        crit.transitions.data = A
        got = crit(em, tgt).detach().cpu().numpy()
        ol, _, _ = O.asg(em.cpu().numpy(), A.cpu().numpy(), tgt.cpu().numpy(), 4)
        err = float(np.abs(got - ol).max() / max(1.0, np.abs(ol).max()))
        check = {"what": "ASG loss [64] on the step's emissions vs fp64 oracle", "max_rel_err": err, "ok": bool(err < 1e-4)}
    return {"config": "conv_glu LibriSpeech ASG (recipes/conv_glu/librispeech/network.arch): B=64/GPU, T=2000, 40 fbank, N=30, fp32",
            "ms_per_step": round(dt * 1e3, 1), "utterances_per_sec": round(B / dt, 2), "finite": bool(torch.isfinite(loss).all().item()),
            "loss_check": check,
            "roofline": {"bound": "mfma", "kernel": "gemm128g_kernel / gemm160_kernel on overlapping-row convolution operands",
                         "achieved": round(tf, 1), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         "launches_per_step": n_.value // steps, "gemm_ms_per_step": round(ms_.value / steps, 1),
                         "algorithmic_tflop_per_step": round(w_.value / steps / 1e12, 2)}}


def ctc_loss_check(tr, x, tgt, what):
    """outside the timed region: the CTC criterion on the network's OWN emissions (eval forward of the stepped
    parameters) through the HIP criterion and through the fp64 oracle, loss [B] element by element (reference call site
    Train.cpp:406-407, :1675; the criterion input is f32 in every precision mode, cpc/Train.cpp:1184)"""
    from oracle import pyoracle as O
    from wav2letter_amd import CTCLoss, CriterionScaleMode
    O.set_num_threads(host_threads())
    em = tr.forward(x, train=False).clone()
    got = CTCLoss(CriterionScaleMode.TARGET_SZ_SQRT)(em, tgt).detach().cpu().numpy()
    want = O.CTC(em.cpu().numpy(), tgt.cpu().numpy(), scale_mode=4).forward()
    fin = np.isfinite(want)
    err = float(np.abs(got[fin] - want[fin]).max() / max(1.0, np.abs(want[fin]).max())) if fin.any() else 0.0
    return {"what": what, "max_rel_err": err, "utterances": int(fin.sum()),
            "ok": bool(err < 1e-4 and (np.isfinite(got) == fin).all())}


def streaming_tds_step(device, L, steps=3, oracle_checks=True):
    """BASELINE config 3 on one GPU: streaming_convnets LibriSpeech TDS-CTC (am_500ms_future_context.arch, 115.1 M
    parameters), batch 64, T = 1500: the full training step in fp32 and in the mixed-precision mode of
    w2l_trainer_set_mixed_precision -- bf16 OPERAND STORAGE for the fl::Linear products (images written once per step,
    gemm_bf16g.hpp) and the TDS convolutions (conv_tds_bf16.hip), fp32 accumulation, master weights, LayerNorm and criterion.
    Parity: tests/test_gpu_trainer.py::test_streaming_tds_config3_* (fp32 and bf16 against the oracle)."""
    import ctypes as C
    from wav2letter_amd import CriterionScaleMode, recipes
    from wav2letter_amd.trainer import Trainer
    B, T, nfeat, nlabel, Lmax = 64, 1500, 80, 9998, 80
    fl = recipes.STREAMING_TDS_FLAGS
    x, tgt = make_batch(B, T, nfeat, nlabel, Lmax, 3, device)
    out = {"config": "streaming_convnets LibriSpeech TDS-CTC (am_500ms_future_context.arch): B=64/GPU, T=1500, 80 mel, 9998 classes"}
    for mode in ("f32", "bf16"):
        tr = Trainer(recipes.streaming_tds_arch(), nfeat, nlabel, "ctc", CriterionScaleMode.TARGET_SZ_SQRT, device=device)
        tr.init_params(seed=1)
        tr.plan(B, T, Lmax)
        tr.to_device()
        tr.set_mixed_precision(mode == "bf16")

        def step():
            loss = tr.forward_backward(x, tgt)
            tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
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
        L.w2l_profile_report_kind(6 if mode == "bf16" else 0, C.byref(n_), C.byref(ms_), C.byref(w_))
        L.w2l_profile_enable(0)
        tf = w_.value / (ms_.value * 1e-3) / 1e12 if ms_.value > 0 else 0.0
        out[mode] = {"ms_per_step": round(dt * 1e3, 1), "utterances_per_sec": round(B / dt, 1),
                     "finite": bool(torch.isfinite(loss).all().item()), "loss": round(float(loss.mean().item()), 4),
                     "gemm_TFLOPs": round(tf, 1), "gemm_ms_per_step": round(ms_.value / steps, 1), "gemm_launches_per_step": n_.value // steps}
        if oracle_checks and mode == "bf16":
            out["loss_check"] = ctc_loss_check(tr, x, tgt, f"CTC loss [{B}] on the {mode} step's emissions (eval forward) vs fp64 oracle")
        del tr
        torch.cuda.empty_cache()
    out["bf16_speedup"] = round(out["f32"]["ms_per_step"] / out["bf16"]["ms_per_step"], 3)
    bf = out["bf16"]
    out["roofline"] = {"bound": "mfma", "kernel": "gemm128h_kernel / gemm256h_kernel (v_mfma_f32_32x32x16_bf16 on bf16 operand images, "
                       "128x128x64 or 256x256x64 tiles, whole-tile persistent schedules)", "achieved": bf["gemm_TFLOPs"],
                       "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(bf["gemm_TFLOPs"] / PEAK_BF16_MFMA_TFLOPS, 4),
                       "covers_frac_of_step_time": round(bf["gemm_ms_per_step"] / bf["ms_per_step"], 4),
                       "hbm_view": "an fc product of this step (M = 11968, N = K = 2160) moves 164 MB of operands + result for 112 GFLOP: "
                                   "680 flop per byte, the HBM bound (8 TB/s) would be 5.4 PFLOP/s -- the GEMMs are matrix-pipe bound, what "
                                   "is HBM-bound in this step is everything else (LayerNorm, image conversions, dropout / masks: about a "
                                   "quarter of the step time, DESIGN 3.1)"}
    out["note"] = ("dtype of this leg: bf16 operands (stored as bf16 images) / fp32 accumulate in the fl::Linear GEMMs and the TDS "
                   "convolutions, fp32 everywhere else")
    return out


def transformer_ctc_step(device, L, steps=3, oracle_checks=True):
    """BASELINE config 5 on one GPU: sota/2019 Transformer-CTC (am_transformer_ctc.arch: WN-conv + GLU + max-pool front end,
    24 blocks of width 1024 / 4 heads / +-460 relative positions, 322.6 M parameters), batch 16, T = 1500 -> 188 frames,
    9998 word pieces: the full training step (dropout 0.2 and layer drop 0.2 live, as the recipe trains) in fp32 and in the
    mixed-precision mode: bf16 operand images for the six fl::Linear of a block (grouped launches for q / k / v and for the four
    projection weight gradients), bf16 MFMA attention -- the fused forward kernel (attention_fused.hip) and the bf16 batched
    GEMM for the backward products.
    Parity: tests/test_gpu_trainer.py::test_transformer_* (incl. the block at the recipe's width in fp32 and bf16),
    tests/test_gpu_attention.py."""
    from wav2letter_amd import CriterionScaleMode, recipes
    from wav2letter_amd.trainer import Trainer
    B, T, nfeat, nlabel, Lmax = 16, 1500, 80, 9998, 80
    fl = recipes.TRANSFORMER_CTC_FLAGS
    x, tgt = make_batch(B, T, nfeat, nlabel, Lmax, 5, device)
    out = {"config": "sota/2019 Transformer-CTC (am_transformer_ctc.arch): B=16/GPU, T=1500 (188 frames after 3 max-pools), 80 mel, 9998 classes"}
    for mode in ("f32", "bf16"):
        tr = Trainer(recipes.transformer_ctc_arch(), nfeat, nlabel, "ctc", CriterionScaleMode.TARGET_SZ_SQRT, device=device)
        tr.init_params(seed=1)
        tr.plan(B, T, Lmax)
        tr.to_device()
        tr.set_mixed_precision(mode == "bf16")
        tr.set_optimizer(fl["netoptim"], fl["critoptim"])
        it = [0]

        def step():
            it[0] += 1
            tr.set_step(it[0])
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
        L.w2l_profile_report_kind(6 if mode == "bf16" else 0, C.byref(n_), C.byref(ms_), C.byref(w_))
        L.w2l_profile_enable(0)
        tf = w_.value / (ms_.value * 1e-3) / 1e12 if ms_.value > 0 else 0.0
        out[mode] = {"ms_per_step": round(dt * 1e3, 1), "utterances_per_sec": round(B / dt, 1),
                     "finite": bool(torch.isfinite(loss).all().item()), "loss": round(float(loss.mean().item()), 4),
                     "gemm_TFLOPs": round(tf, 1), "gemm_ms_per_step": round(ms_.value / steps, 1), "gemm_launches_per_step": n_.value // steps}
        if oracle_checks and mode == "bf16":
            out["loss_check"] = ctc_loss_check(tr, x, tgt, f"CTC loss [{B}] on the {mode} step's emissions (eval forward) vs fp64 oracle")
        del tr
        torch.cuda.empty_cache()
    out["bf16_speedup"] = round(out["f32"]["ms_per_step"] / out["bf16"]["ms_per_step"], 3)
    bf = out["bf16"]
    out["roofline"] = {"bound": "mfma", "kernel": "gemm128h_kernel (incl. its grouped launches)", "achieved": bf["gemm_TFLOPs"],
                       "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(bf["gemm_TFLOPs"] / PEAK_BF16_MFMA_TFLOPS, 4),
                       "covers_frac_of_step_time": round(bf["gemm_ms_per_step"] / bf["ms_per_step"], 4),
                       "note": "M = 3008 frames per step: 192-tile grids and 16-K-tile reductions on a 256-CU chip (DESIGN 3.1)"}
    out["note"] = ("optimizer: the recipe's --netoptim=adadelta (librispeech/train_am_transformer_ctc.cfg); layer drop skips a dropped "
                   "block's GEMMs, so ms_per_step is the mean over the masks these steps drew")
    return out


def asg_stress(device, L, T, oracle_checks=True):
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

    def kind(k):
        n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
        L.w2l_profile_report_kind(k, C.byref(n_), C.byref(ms_), C.byref(w_))
        return n_.value, ms_.value, w_.value
    L.w2l_profile_enable(1)
    t0 = time.perf_counter()
    loss = crit(x, tgt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nl, ms, by = kind(3)             # the alpha pass's transition-stream launches
    L.w2l_profile_enable(1)          # (resets the event list)
    loss.sum().backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    nlb, msb, byb = kind(3)          # the beta pass's transition-stream launches
    nda, msda, flda = kind(0)        # dA = (g r)^T e: one fp32 MFMA GEMM over (t, b), not part of the HBM-bound recursion
    L.w2l_profile_enable(0)
    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    check = None
    if oracle_checks:
        # outside the timed region: utterance 0 of the TIMED forward (fp32 exp-domain rescaling over T dependent steps at
        # N = 9998) against the fp64 log-domain oracle: 1.5e11 log-sum-exp terms on the host cores, label loop threaded
        # The oracle costs N^2 log-sum-exp terms per frame: it is calibrated on 3 frames and then given the first Tc
        # frames that fit ~12 s of host time (the full T = 1500 recursion against fp64 is a -m gpu test at N = 1000:
        # tests/test_gpu_parity_shapes.py::test_fcc_long_recursion_t1500_matches_fp64_oracle)
        from oracle import pyoracle as O
        An = crit.transitions.detach().cpu().numpy()
        x0 = x[0:1].detach().cpu().numpy()
        O.set_num_threads(host_threads())
        t3 = time.perf_counter()
        O.FCC(np.ascontiguousarray(x0[:, :9]), An, np.array([8], np.int32), 4).forward()
        per_frame = max(1e-4, (time.perf_counter() - t3) / 8)
        Tc = int(max(8, min(T, 8.0 / per_frame)))
        note(f"stress oracle check: {per_frame * 1e3:.0f} ms per frame on the host -> first {Tc} frames")
        xc = np.ascontiguousarray(x0[:, :Tc])
        want = float(O.FCC(xc, An, np.array([8], np.int32), 4).forward()[0])
        with torch.no_grad():
            got = float(crit(torch.from_numpy(xc).to(device), tgt[0:1])[0].item())
        check = {"what": f"FCC loss of utterance 0 over its first {Tc} frames (N={N}, random transitions) vs fp64 oracle",
                 "frames": Tc, "got": got, "oracle": want, "rel_err": abs(got - want) / max(1.0, abs(want)),
                 "ok": bool(abs(got - want) < 1e-4 * max(1.0, abs(want))), "oracle_seconds": round(time.perf_counter() - t3, 1)}
    step_bytes = 4.0 * N * N + 8.0 * B * N
    fwd_ms, bwd_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
    whole = step_bytes * (T - 1) / (fwd_ms * 1e-3) / 1e9
    achieved_b = byb / (msb * 1e-3) / 1e9 if msb > 0 else 0.0
    beta_ms = bwd_ms - msda          # the recursion proper: the dA GEMM (MFMA-bound, 2 N^2 T B flop) is timed by its own events
    whole_b = step_bytes * (T - 1) / (beta_ms * 1e-3) / 1e9 if beta_ms > 0 else 0.0
    tr_, trd = pmc_traffic("fcc_big_gemm")
    return {"shape": f"B={B},T={T},N={N}", "fwd_ms": round(fwd_ms, 2), "bwd_ms": round(bwd_ms, 2),
            "fwd_us_per_step": round(fwd_ms * 1e3 / T, 2),
            "whole_forward_GBps": round(whole, 1),
            "finite": bool(torch.isfinite(loss).all().item()), "loss_check": check,
            # frac = the WHOLE alpha pass (every launch of the T dependent steps, wall time); the streaming kernel alone beside it
            "roofline": {"bound": "hbm", "kernel": "alpha pass of FullConnectionCriterion at N = 9998: per frame fcc_big_gemm_dma "
                                                   "(packed-transition stream, fp32 MFMA 32x32x2) + fcc_big_step",
                         "achieved": round(whole, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(whole / PEAK_HBM_GBPS, 4), "traffic": tr_, "traffic_detail": trd,
                         "algorithmic_bytes_per_frame": step_bytes, "frames": T - 1,
                         "stream_kernel_only": {"achieved": round(achieved, 1), "frac": round(achieved / PEAK_HBM_GBPS, 4),
                                                "launches": nl, "avg_launch_us": round(ms * 1e3 / max(1, nl), 2)},
                         "beta_pass": {"achieved": round(whole_b, 1), "frac": round(whole_b / PEAK_HBM_GBPS, 4),
                                       "ms": round(beta_ms, 2), "note": "bwd_ms minus the dA GEMM",
                                       "dA_gemm_ms": round(msda, 2), "dA_gemm_TFLOPs": round(flda / (msda * 1e-3) / 1e12, 1) if msda > 0 else None,
                                       "stream_kernel_only": {"achieved": round(achieved_b, 1), "frac": round(achieved_b / PEAK_HBM_GBPS, 4),
                                                              "launches": nlb, "avg_launch_us": round(msb * 1e3 / max(1, nlb), 2)}}}}


def host_threads():
    """threads for the host-side legs: the cores this process may use, capped at 64 -- 256 OpenMP threads on the GPU
    box's 256 logical CPUs ran the torch-CPU step 50x SLOWER than 8 threads on an 8-core host (round-2 run B: 122 s per
    utterance), two OpenMP runtimes (torch's and the oracle's) spinning against each other"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def cpu_baseline(nfeat, nlabel, T, batch=8):
    """The reference's CPU path as BASELINE.md sec. 3 item 2 / SURVEY 8(d) prescribe it -- a declared PROXY, the
    reference (Flashlight + ArrayFire) cannot be built here: torch-CPU (oneDNN / MKL, the libraries Flashlight's CPU
    backend sits on) with every host core for conv / linear / LayerNorm, the OpenMP oracle for CTC; one training step
    (forward + CTC + backward, no optimizer) of the SAME arch at the SAME T on `batch` utterances, 2 warm-ups, median of
    up to 5 runs, bounded to ~30 s of CPU work.  A reported baseline, not the target."""
    from oracle import pyoracle as O
    from oracle import torchnet
    from wav2letter_amd import recipes
    cores = host_threads()
    torch.set_num_threads(cores)
    O.set_num_threads(cores)
    os.environ["OMP_WAIT_POLICY"] = "PASSIVE"
    t0 = time.perf_counter()
    # size the sample: a one-utterance probe step, then the batch and as many timed runs as fit ~30 s
    probe8, _ = torchnet.tds_ctc_step_seconds(recipes.tds_ctc_arch(), nfeat, nlabel, 1, max(64, T // 8), warmup=0, runs=1)
    note(f"cpu baseline pre-probe: 1 utterance x {max(64, T // 8)} frames in {probe8:.2f} s on {cores} threads")
    if probe8 > 6.0:   # a full-length utterance would blow the budget: report the short sample, scaled
        return {"value": round((max(64, T // 8) / T) / probe8, 4), "unit": "utterances/sec", "cores": cores,
                "kind": "port", "port_of": "oracle/torchnet.py (torch-CPU oneDNN/MKL restatement of the network) + oracle CTC (OpenMP C)", "sample": f"1 utterance x {max(64, T // 8)} frames scaled to T={T} (single run, {probe8:.1f} s): "
                "the host is too slow for the full sample inside the 30 s budget"}
    probe, _ = torchnet.tds_ctc_step_seconds(recipes.tds_ctc_arch(), nfeat, nlabel, 1, T, warmup=0, runs=1)
    note(f"cpu baseline probe: 1 utterance in {probe:.2f} s on {cores} threads")
    batch = int(max(1, min(batch, 6.0 / max(probe, 1e-3))))       # a step of `batch` utterances within ~6 s
    est = probe * max(1.0, batch * 0.6)
    runs = int(max(1, min(5, 24.0 / max(est, 1e-3) - 2)))
    warm = 2 if runs >= 3 else 1
    med, times = torchnet.tds_ctc_step_seconds(recipes.tds_ctc_arch(), nfeat, nlabel, batch, T, warmup=warm, runs=runs)
    return {"value": round(batch / med, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "port_of": "oracle/torchnet.py (torch-CPU oneDNN/MKL restatement of the network) + oracle CTC (OpenMP C)",
            "torch_threads": torch.get_num_threads(), "oracle_threads": O.num_threads(),
            "sample": f"{batch} utterances x {T} frames, TDS-CTC training step (forward + CTC + backward, no optimizer), "
                      f"torch-CPU oneDNN/MKL network + OpenMP oracle CTC; {warm} warm-up(s), median of {runs} run(s) "
                      f"({', '.join(f'{t:.2f}' for t in times)} s); {time.perf_counter() - t0:.1f} s of CPU work"}


def input_pipeline_leg(device, tr, tgt, fl, B, T, nfeat, ms_step_resident, steps=10):
    """SURVEY 8 f3 'input pipeline at speed': the prefetching loader (wav2letter_amd/loader.py: decode threads -> pinned ring
    -> H2D on a side stream -> MFSC on the device) timed alone and underneath the headline training step.  Synthetic 16-bit
    mono WAV files of exactly T frames (15 s at T = 1500), written once to a temporary directory and re-read every batch
    (page-cache resident: this measures decode + pad + PCIe + MFSC, not a disk)."""
    import shutil
    import tempfile
    import wave
    import numpy as np
    from wav2letter_amd.features import Mfsc
    from wav2letter_amd.loader import PrefetchLoader, read_audio_int16
    mfsc = Mfsc(num_filters=nfeat, device=device)
    ns = (T - 1) * mfsc.S + mfsc.N
    tmp = tempfile.mkdtemp(prefix="w2l_bench_wav_")
    try:
        rng = np.random.default_rng(5)
        paths = []
        for i in range(2 * B):
            pth = os.path.join(tmp, f"u{i:03d}.wav")
            with wave.open(pth, "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
                w.writeframes((rng.normal(size=ns) * 3000).astype("<i2").tobytes())
            paths.append(pth)
        nb = steps + 2
        batches = [[(n * B + j) % len(paths) for j in range(B)] for n in range(nb)]
        workers = min(8, host_threads())
        mk = lambda: PrefetchLoader(paths, batches, mfsc, device=device, workers=workers, depth=3, read=read_audio_int16)
        # alone
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for feats, sizes, _ids in mk():
            n += feats.shape[0]
        torch.cuda.synchronize()
        alone = time.perf_counter() - t0
        assert tuple(feats.shape) == (B, nfeat, T), tuple(feats.shape)
        # underneath the training step (2 untimed batches first)
        it = iter(mk())
        for _ in range(2):
            feats, sizes, _ids = next(it)
            tr.forward_backward(feats, tgt)
            tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for feats, sizes, _ids in it:
            loss = tr.forward_backward(feats, tgt)
            tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
            k += 1
        torch.cuda.synchronize()
        fed = (time.perf_counter() - t0) / max(1, k)
        # the same through the library's FLAC decoder (csrc/host/flac.cpp: frame CRCs + STREAMINFO MD5 verified per file): four
        # files written by the tests' specification-level encoder (LPC order 8, the shape of a libFLAC level-5 stream)
        flac = None
        try:
            from tests import flac_encode as FE
            fpaths = []
            for i in range(4):
                with wave.open(paths[i], "rb") as w_:
                    pcm = np.frombuffer(w_.readframes(ns), "<i2").astype(np.int64)
                sm = np.convolve(pcm, np.ones(8) / 8.0, mode="same").round().astype(np.int64)   # low-passed noise: predictable
                fp = os.path.join(tmp, f"f{i}.flac")
                with open(fp, "wb") as f_:
                    f_.write(FE.encode(sm, kind="lpc", order=8, porder=4, lpc=(12, 9, [970, -300, 120, -60, 30, -10, 5, -2])))
                fpaths.append(fp)
            fb = [[(n_ * B + j) % 4 for j in range(B)] for n_ in range(6)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nf = 0
            for feats_f, _s, _i in PrefetchLoader(fpaths, fb, mfsc, device=device, workers=workers, depth=3, read=read_audio_int16):
                nf += feats_f.shape[0]
            torch.cuda.synchronize()
            tf = time.perf_counter() - t0
            flac = {"utterances_per_sec": round(nf / tf, 1), "compressed_fraction_of_pcm": round(os.path.getsize(fpaths[0]) / (2.0 * ns), 3)}
        except Exception as e:  # noqa: BLE001
            flac = {"error": f"{type(e).__name__}: {e}"}
        return {"loader_alone": {"utterances_per_sec": round(n / alone, 1), "ms_per_batch": round(alone / nb * 1e3, 2),
                                 "pcm_MB_per_sec": round(n * ns * 2 / alone / 1e6, 1)},
                "loader_alone_flac": flac,
                "step_fed_by_loader": {"ms_per_step": round(fed * 1e3, 3), "utterances_per_sec": round(B / fed, 2), "steps": k,
                                       "vs_resident_input": round(fed * 1e3 / ms_step_resident, 4),
                                       "finite": bool(torch.isfinite(loss).all().item())},
                "needed_utterances_per_sec": round(B / (ms_step_resident * 1e-3), 1),
                "workers": workers, "depth": 3,
                "source": f"{len(paths)} synthetic 16-bit mono WAV files x {ns} samples ({ns / 16000:.2f} s), page-cache resident; "
                          "int16 over PCIe, scaling + MFSC (two GEMMs) on the device, side stream"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dist_selftest(a):
    """launcher + rendezvous + the arena all-reduce, nothing else: every rank fills a flat gradient arena whose tail
    carries its local batch size (parallel.GradientArena, the layout Trainer.grads_full has), ONE all-reduce sums both.
    Runs with gloo on a CPU-only host (tests/test_distributed_cpu.py drives it through the same self-launch path as a
    real --gpus N run) and with RCCL when GPUs are visible."""
    import torch.distributed as dist
    from wav2letter_amd.parallel import GradientArena, init_distributed
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local_rank)
    rank, world = init_distributed("nccl" if use_gpu else "gloo", device if use_gpu else None)
    n = 1000
    arena = GradientArena(n, device)
    arena.grads.fill_(float(rank + 1))
    arena.set_local_batch(a.batch + rank)       # ragged tail batches: ranks may differ
    total = arena.all_reduce()
    want_g = world * (world + 1) / 2
    want_b = world * a.batch + world * (world - 1) / 2
    ok = bool((arena.grads == want_g).all().item()) and float(total.item()) == want_b
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "dist", "ok": ok, "n_gpus": world, "rccl_ranks": ranks,
                          "backend": "nccl(rccl)" if use_gpu else "gloo", "global_batch": float(total.item())}), flush=True)
    return 0 if ok else 1


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL over xGMI)
        from wav2letter_amd.parallel import self_launch
        sys.exit(self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:],
                             need_gpus=not (a.dist_selftest and not torch.cuda.is_available())))
    if a.dist_selftest:
        sys.exit(dist_selftest(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        # a launcher that started a different number of ranks than --gpus asks for would report a throughput for the
        # wrong N: refuse (the driver launches `torch.distributed.run --nproc-per-node N bench.py --gpus N`)
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the launcher and --gpus disagree")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, this node shows {torch.cuda.device_count()}")
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

    reducer = None
    if dist is not None:
        from wav2letter_amd.parallel import OverlappedReducer
        reducer = OverlappedReducer(tr, n_buckets=a.buckets)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
        if rank == 0:
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                ver = f"unknown ({type(e).__name__})"
            note(f"data parallel: {dist.get_world_size()} ranks over RCCL {ver} (torch backend 'nccl'), "
                 f"{len(reducer.bucket_bytes())} gradient buckets per step, bytes in issue order (last layers first): "
                 f"{reducer.bucket_bytes()}; arena {4 * tr.grads_full.numel()} bytes incl. the 16-byte batch-size tail")

    def step():
        loss = tr.forward_backward(x, tgt)
        if reducer is not None:
            # the flat gradient arena (814 MB fp32) in a few large buckets, last layers first, on a side
            # stream gated by per-bucket events: the sum crosses xGMI under the rest of the backward pass
            reducer.reduce()
        # gradients / (all-reduced batch size): the local batch size rides in the arena's tail through the SAME
        # collective (the reference all-reduces it separately, Train.cpp:1743-1747); world == 1: the local B
        tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"],
                  total_batch="reduced" if reducer is not None else B)
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
    note(f"headline: {a.steps} steps in {dt:.2f} s")
    last_loss = float(loss.float().mean().item())
    # the global batch the optimizer divided by: read back from the (all-reduced) arena tail, not assumed
    total_batch = int(round(tr.grads_full[tr.n_floats].item())) if reducer is not None else B
    rccl_ranks = dist.get_world_size() if dist is not None else 1
    skipped = tr.skipped_updates()
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
    nbd, msbd, flbd = kind(4)        # TDS convolution backward-data
    nbf, msbf, flbf = kind(5)        # TDS convolution backward-filter
    L.w2l_profile_enable(0)
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    step_flops = (flops + flc + flbd + flbf + fls) / max(1, a.steps)   # every MFMA launch of the step, algorithmic 2MNK
    ms_step = dt / a.steps * 1e3

    def tfs(f_, m_):
        return round(f_ / (m_ * 1e-3) / 1e12, 2) if m_ > 0 else None
    out = {
        "metric": "utterances/sec", "value": round(total_batch * a.steps / dt, 3), "unit": "utterances/sec",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "seq2seq_tds LibriSpeech TDS-CTC (sota/2019 am_tds_ctc.arch): 80-mel x T=%d, "
                               "9998 classes, batch %d/GPU, fp32, SGD+momentum, CTC" % (T, B),
                   "global_batch": total_batch, "frames": T, "emission_frames": Tout, "parallelism": f"dp{world}",
                   "params": int(tr.n_net), "final_loss": round(last_loss, 4), "skipped_updates": int(skipped)},
        "roofline": {"bound": "mfma", "kernel": "gemm128g_kernel / gemm160_kernel (fp32 v_mfma_f32_32x32x2_f32; 128x128x32 tiles, or 128x160 / 160x128 where 128 leaves a ragged tile column; persistent, buffer LDS-DMA staging, stream-K tail reduced in-kernel; includes the few gemm128_kernel launches on unaligned shapes)",
                     "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic("gemm_lds_dma", "gemm128g")[0],
                     "traffic_detail": pmc_traffic("gemm_lds_dma", "gemm128g")[1],
                     "launches_per_step": nl // max(1, a.steps),
                     "avg_launch_us": round(ms * 1e3 / max(1, nl), 1),
                     "algorithmic_gflop_per_launch": round(flops / max(1, nl) / 1e9, 2),
                     "gemm_ms_per_step": round(ms / max(1, a.steps), 3),
                     "algorithmic_tflop_per_step": round(flops / max(1, a.steps) / 1e12, 3),
                     "covers_frac_of_step_time": round(ms / max(1, a.steps) / ms_step, 4),
                     "whole_step": {"algorithmic_tflop": round(step_flops / 1e12, 3), "achieved_TFLOPs": round(step_flops / (ms_step * 1e-3) / 1e12, 2),
                                    "frac": round(step_flops / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                    "note": "all MFMA work of the step (GEMMs + convolutions) / wall time per step: what the HBM-bound "
                                            "LayerNorm / dropout / CTC / optimizer passes and launch gaps cost on top of the dominant kernel"},
                     "tds_conv": {"launches_per_step": (nc + nbd + nbf) // max(1, a.steps),
                                  "ms_per_step": round((msc + msbd + msbf) / max(1, a.steps), 3),
                                  "achieved_TFLOPs": tfs(flc + flbd + flbf, msc + msbd + msbf),
                                  "forward": {"launches_per_step": nc // max(1, a.steps), "ms_per_step": round(msc / max(1, a.steps), 3), "achieved_TFLOPs": tfs(flc, msc)},
                                  "backward_data": {"launches_per_step": nbd // max(1, a.steps), "ms_per_step": round(msbd / max(1, a.steps), 3), "achieved_TFLOPs": tfs(flbd, msbd)},
                                  "backward_filter": {"launches_per_step": nbf // max(1, a.steps), "ms_per_step": round(msbf / max(1, a.steps), 3), "achieved_TFLOPs": tfs(flbf, msbf)},
                                  "note": "TDS convolutions proper: wave-specialised role-swapped v_mfma_f32_32x32x2_f32 kernels (conv_tds_rs3.hpp at C = 10 / 14 / 18, "
                                          "conv_tds_rsf3.hpp at C = 10 / 18, conv_tds.hip's filter gradient at C = 14); includes the three strided C2 sub-sampling layers"},
                     "skinny_gemm": {"launches_per_step": ns // max(1, a.steps), "ms_per_step": round(mss / max(1, a.steps), 3)}},
    }
    def leg(key, fn):
        # the additional legs never take the headline line down with them: a failure is reported in place
        note(f"leg {key} ...")
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    if world == 1 and not a.no_input_pipeline:
        leg("input_pipeline", lambda: input_pipeline_leg(device, tr, tgt, fl, B, T, nfeat, ms_step))
    if not a.no_asg:
        leg("asg_loss_ms_per_step", lambda: asg_criterion_ms(device))
        if world == 1:
            leg("ctc_loss_ms_per_step", lambda: ctc_criterion_ms(device, L))
    if world == 1 and not a.no_stress:
        del tr, x, tgt
        torch.cuda.empty_cache()
        leg("asg_stress", lambda: asg_stress(device, L, a.stress_frames, not a.no_oracle_checks))
    if world == 1 and not a.no_c4:
        leg("conv_glu_asg_step", lambda: conv_glu_asg_step(device, L, oracle_checks=not a.no_oracle_checks))
    if world == 1 and not a.no_c3:
        leg("streaming_tds_bf16_step", lambda: streaming_tds_step(device, L, oracle_checks=not a.no_oracle_checks))
    if world == 1 and not a.no_c5:
        leg("transformer_ctc_step", lambda: transformer_ctc_step(device, L, oracle_checks=not a.no_oracle_checks))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if world == 1 and not a.no_cpu_baseline:
        note("leg cpu_baseline ...")
        try:
            out["cpu_baseline"] = cpu_baseline(nfeat, nlabel, T, a.cpu_baseline_batch)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    note("done")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
