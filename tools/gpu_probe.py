#!/usr/bin/env python
"""GPU micro-benchmarks of the individual hot-path kernels (run on the MI355X box via gpurun).
Prints one line per measurement; used to fill DESIGN.md's per-kernel roofline table.

  python tools/gpu_probe.py gemm ln conv asg fccbig vitbig
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from wav2letter_amd import _lib, ops
from wav2letter_amd.criterion import (ASGLoss, CriterionScaleMode, ForceAlignmentCriterion, FullConnectionCriterion)


def timeit(fn, n=10, warm=2):
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


def probe_gemm():
    shapes = [("fc1 s1", 24000, 800, 2400), ("fc2 s1", 24000, 2400, 800), ("fc1 s2", 12000, 1120, 3360),
              ("fc2 s2", 12000, 3360, 1120), ("fc1 s3", 6016, 1440, 4320), ("fc2 s3", 6016, 4320, 1440),
              ("final", 6016, 1440, 9998), ("4096^3", 4096, 4096, 4096)]
    for name, M, K, N in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        dx = torch.empty(M, K, device="cuda")
        dw = torch.empty(K, N, device="cuda")
        L = _lib.lib()
        s = torch.cuda.current_stream().cuda_stream
        fl = 2.0 * M * N * K
        for sk in ("1", "0"):
            os.environ["W2L_GEMM_SK"] = sk
            tf = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s))
            td = timeit(lambda: L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s))
            tw = timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))
            print(f"[gemm] {name:7s} M={M} K={K} N={N} sk={sk}: fwd {tf:.3f} ms {fl / tf / 1e9:.1f} TF | "
                  f"dX {td:.3f} ms {fl / td / 1e9:.1f} TF | dW {tw:.3f} ms {fl / tw / 1e9:.1f} TF", flush=True)
        os.environ["W2L_GEMM_SK"] = "1"


def probe_gemm160():
    """TDS fc shapes: 128x128 tile (W2L_GEMM_T160=0) against the padded-area rule (default) -- after a burn-in, the
    first measurement of a process runs at ramping clocks (profiles/r01_run30_gemm_offsets.log)"""
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    a = torch.randn(4096, 4096, device="cuda")
    c = torch.empty(4096, 4096, device="cuda")
    timeit(lambda: L.w2l_linear_forward(4096, 4096, 4096, a.data_ptr(), a.data_ptr(), None, c.data_ptr(), 0, s), n=60)
    shapes = [("fc1 s1", 24000, 800, 2400), ("fc2 s1", 24000, 2400, 800), ("fc1 s2", 12000, 1120, 3360),
              ("fc2 s2", 12000, 3360, 1120), ("fc1 s3", 6016, 1440, 4320), ("fc2 s3", 6016, 4320, 1440), ("4096^3", 4096, 4096, 4096)]
    tot = {"0": 0.0, "1": 0.0, "2": 0.0, "3": 0.0}
    for name, M, K, N in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        dx = torch.empty(M, K, device="cuda")
        dw = torch.empty(K, N, device="cuda")
        fl = 2.0 * M * N * K
        for mode in (os.environ.get("PROBE_T160_MODES", "0,1,0,1").split(",")):
            os.environ["W2L_GEMM_T160"] = mode
            tf = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s), n=20)
            td = timeit(lambda: L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s), n=20)
            tw = timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s), n=20)
            tot[mode] += tf + td + tw
            print(f"[gemm160 t160={mode}] {name:7s} M={M} K={K} N={N}: fwd {tf * 1e3:.0f} us {fl / tf / 1e9:.1f} TF | "
                  f"dX {td * 1e3:.0f} us {fl / td / 1e9:.1f} TF | dW {tw * 1e3:.0f} us {fl / tw / 1e9:.1f} TF", flush=True)
    os.environ.pop("W2L_GEMM_T160")
    print(f"[gemm160] sum over shapes (two passes each): 128x128 {tot['0']:.2f} ms, padded-area rule {tot['1']:.2f} ms", flush=True)


def probe_gemmfwd():
    """forward GEMM only (ablation runs: W2L_GEMM_ABL is read once per process)"""
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for name, M, K, N in [("fc1 s1", 24000, 800, 2400), ("fc2 s3", 6016, 4320, 1440), ("4096^3", 4096, 4096, 4096),
                          ("8192^3", 8192, 8192, 8192)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        fl = 2.0 * M * N * K
        tf = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s), n=20, warm=3)
        print(f"[gemmfwd abl={os.environ.get('W2L_GEMM_ABL', '0')} glds={os.environ.get('W2L_GEMM_GLDS', '1')}] "
              f"{name:7s} M={M} K={K} N={N}: {tf:.3f} ms {fl / tf / 1e9:.1f} TF", flush=True)


def probe_ln():
    for name, B, inner in [("tds s1", 32, 750 * 800), ("tds s2", 32, 375 * 1120), ("tds s3", 32, 188 * 1440),
                           ("frame", 32 * 188, 1440)]:
        a = torch.randn(B, inner, device="cuda").clamp_min(0)
        x = torch.randn(B, inner, device="cuda")
        gb = torch.tensor([1.1, 0.1], device="cuda")
        dy = torch.randn(B, inner, device="cuda")
        nbytes = B * inner * 4
        y, r, mr = ops.residual_layernorm_forward(a.clone(), x, gb, B, 1e-5, 0.2, 3, 1)
        tf = timeit(lambda: ops.residual_layernorm_forward(a, x, gb, B, 1e-5, 0.2, 3, 1))
        tb = timeit(lambda: ops.layernorm_backward(r, dy, gb, mr, B, mask_src=a, mask_scale=1.25))
        # fwd: read a,x write a,r (stats pass) + read r write y ; bwd: read r,dy twice, read mask, write dr,dmask
        print(f"[ln] {name}: B={B} inner={inner} fwd {tf * 1e3:.1f} us ({6 * nbytes / tf / 1e9:.2f} TB/s alg) "
              f"bwd {tb * 1e3:.1f} us ({7 * nbytes / tb / 1e9:.2f} TB/s alg)  [includes torch.empty allocs]", flush=True)


def probe_conv():
    L = _lib.lib()
    import ctypes as C
    s = torch.cuda.current_stream().cuda_stream
    for name, B, T, H, Cc, kw in [("tds s1", 32, 750, 80, 10, 21), ("tds s2", 32, 375, 80, 14, 21), ("tds s3", 32, 188, 80, 18, 21)]:
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        x = torch.randn(B, T, H, Cc, device="cuda")
        w = torch.randn(kw, Cc, Cc, device="cuda")
        b = torch.randn(Cc, device="cuda")
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        fl = 2.0 * B * T * H * Cc * Cc * kw
        nb = x.numel() * 4
        tf = timeit(lambda: L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s))
        td = timeit(lambda: L.w2l_conv_backward_data(C.byref(d), y.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, s))
        tw = timeit(lambda: L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), s))
        print(f"[conv] {name}: fwd {tf * 1e3:.0f} us {fl / tf / 1e9:.1f} TF {2 * nb / tf / 1e9:.2f} TB/s | "
              f"dX {td * 1e3:.0f} us {fl / td / 1e9:.1f} TF | dW+db {tw * 1e3:.0f} us {fl / tw / 1e9:.1f} TF", flush=True)


def probe_convglu():
    """conv_glu LibriSpeech (BASELINE config C4) WN-conv layers at B=64, T=2000: first / middle / last layer shapes of
    recipes/conv_glu/librispeech/network.arch (valid convolutions, GLU halves the channels in between)"""
    L = _lib.lib()
    import ctypes as C
    s = torch.cuda.current_stream().cuda_stream
    B, T = 64, 2000
    for name, Cin, Cout, kw in [("L1 40->400 k13", 40, 400, 13), ("L5 266->584 k17", 266, 584, 17), ("L9 388->852 k21", 388, 852, 21),
                                ("L13 565->1242 k25", 565, 1242, 25), ("L17 826->1816 k29", 826, 1816, 29)]:
        d = _lib.ConvDesc(B, T, 1, Cin, Cout, kw, 1, 0, 0)
        To = T - kw + 1
        x = torch.randn(B, T, 1, Cin, device="cuda")
        w = torch.randn(kw, Cin, Cout, device="cuda") / (kw * Cin) ** 0.5
        b = torch.randn(Cout, device="cuda")
        y = torch.empty(B, To, 1, Cout, device="cuda")
        dy = torch.randn(B, To, 1, Cout, device="cuda")
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        fl = 2.0 * B * To * Cin * Cout * kw
        tf = timeit(lambda: L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, s), n=3, warm=1)
        td = timeit(lambda: L.w2l_conv_backward_data(C.byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, s), n=3, warm=1)
        tw = timeit(lambda: L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s), n=3, warm=1)
        print(f"[convglu] {name} ({fl / 1e12:.2f} TFLOP): fwd {tf:.2f} ms {fl / tf / 1e9:.1f} TF | dX {td:.2f} ms {fl / td / 1e9:.1f} TF | "
              f"dW+db {tw:.2f} ms {fl / tw / 1e9:.1f} TF", flush=True)
        del x, w, y, dy, dx, dw


def probe_feat():
    """log-mel front end at the TDS-CTC batch shape: 32 utterances of 15 s at 16 kHz -> [32][80][1498]"""
    from wav2letter_amd.features import Mfsc
    B, ns = 32, 240000
    audio = torch.randn(B, ns, device="cuda") * 3000
    fe = Mfsc(80)
    T = fe.num_frames(ns)
    t = timeit(lambda: fe(audio), n=20, warm=3)
    fl = 2.0 * B * T * (fe.N * 2 * fe.nb + fe.ld * fe.F)
    print(f"[feat] MFSC B={B} x {ns / 16000:.0f} s -> [{B}][80][{T}]: {t * 1e3:.0f} us = {B / t * 1e3:.0f} utterances/s, "
          f"{B * ns / 16000 / (t * 1e-3):.0f}x real time, {fl / t / 1e9:.1f} TF/s on the two GEMMs' {fl / 1e9:.1f} GFLOP", flush=True)


def asg_targets(B, L, g):
    tgt = torch.full((B, L), -1, dtype=torch.int32)
    for b in range(B):
        l = int(torch.randint(60, L + 1, (1,), generator=g))
        y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
        for i in range(1, l):
            if y[i] == y[i - 1]:
                y[i] = (y[i] + 1) % 28
        tgt[b, :l] = y
    return tgt


def probe_asg():
    B, T, N, L = 64, 2000, 30, 300
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, T, N, generator=g).cuda()
    tgt = asg_targets(B, L, g).cuda()
    A = (torch.eye(N) * 4 + torch.randn(N, N, generator=g) * 0.1).cuda()
    for name, cls in [("FCC", FullConnectionCriterion), ("FAC", ForceAlignmentCriterion)]:
        crit = cls(N, CriterionScaleMode.TARGET_SZ_SQRT).cuda()
        crit.transitions.data = A.clone()
        xr = x.clone().requires_grad_(True)
        tf = timeit(lambda: crit(xr, tgt))
        tfb = timeit(lambda: crit(xr, tgt).sum().backward())
        print(f"[asg] {name} B={B} T={T} N={N}: fwd {tf:.3f} ms, fwd+bwd {tfb:.3f} ms", flush=True)
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).cuda()
    tv = timeit(lambda: crit.viterbiPath(x))
    print(f"[asg] Viterbi B={B} T={T} N={N}: {tv:.3f} ms", flush=True)


def probe_fccbig(T=64):
    B, N = 32, 9998
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, T, N, generator=g).cuda()
    A = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
    tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
    crit = FullConnectionCriterion(N, CriterionScaleMode.NONE).cuda()
    crit.transitions.data = A
    xr = x.requires_grad_(True)
    tf = timeit(lambda: crit(xr, tgt), n=3, warm=1)
    tfb = timeit(lambda: crit(xr, tgt).sum().backward(), n=3, warm=1)
    step_bytes = 4.0 * N * N + 8.0 * B * N
    print(f"[fccbig] B={B} T={T} N={N}: fwd {tf:.2f} ms = {tf / T * 1e3:.1f} us/step "
          f"({step_bytes * (T - 1) / tf / 1e9:.2f} TB/s algorithmic incl. packing), fwd+bwd {tfb:.2f} ms", flush=True)


def probe_fccstream(T=200):
    """per-launch time of the N=9998 transition stream (fcc_big_gemm) by the library's own HIP events"""
    import ctypes as C
    B, N = 32, 9998
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, T, N, generator=g).cuda().requires_grad_(True)
    tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
    crit = FullConnectionCriterion(N, CriterionScaleMode.NONE).cuda()
    crit.transitions.data = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
    crit(x, tgt).sum().backward()
    torch.cuda.synchronize()
    L = _lib.lib()
    L.w2l_profile_enable(1)
    crit(x, tgt).sum().backward()
    torch.cuda.synchronize()
    nl, ms, by = C.c_int(0), C.c_double(0), C.c_double(0)
    L.w2l_profile_report_kind(3, C.byref(nl), C.byref(ms), C.byref(by))
    L.w2l_profile_enable(0)
    env = {k: os.environ[k] for k in ("W2L_FCC_RT", "W2L_FCC_RING", "W2L_FCC_ABL", "W2L_FCC_WPC") if k in os.environ}
    print(f"[fccstream {env}] {nl.value} launches, {ms.value * 1e3 / nl.value:.2f} us/launch, "
          f"{by.value / (ms.value * 1e-3) / 1e12:.3f} TB/s algorithmic", flush=True)


def probe_vitbig(T=24):
    N = 9998
    g = torch.Generator(device="cpu").manual_seed(6)
    A = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
    crit = ASGLoss(N, CriterionScaleMode.NONE, 0.0).cuda()
    crit.transitions.data = A
    for B in (1, 32):
        x = torch.randn(B, T, N, generator=g).cuda()
        tv = timeit(lambda: crit.viterbiPath(x), n=2, warm=1)
        print(f"[vitbig] B={B} T={T} N={N}: {tv:.2f} ms = {tv / T * 1e3:.1f} us/step "
              f"({4.0 * N * N * (T - 1) / tv / 1e9:.2f} TB/s algorithmic)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "ln", "conv", "asg", "fccbig", "vitbig"]
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for w in which:
        t0 = time.time()
        {"gemm": probe_gemm, "gemm160": probe_gemm160, "gemmfwd": probe_gemmfwd, "ln": probe_ln, "conv": probe_conv, "convglu": probe_convglu, "feat": probe_feat, "asg": probe_asg, "fccbig": probe_fccbig, "fccstream": probe_fccstream,
         "vitbig": probe_vitbig}[w]()
        print(f"[{w}] done in {time.time() - t0:.1f} s", flush=True)
