"""bf16 GEMM with k-major operands read in place (w2l_gemm_bf16_ex) against the k-contiguous kernel on transposed images, at the
config-3 / config-5 shapes of the three products of an fl::Linear.   python tools/gemm_kmajor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("c3 wgrad 1200x1200x11968", 1200, 1200, 11968), ("c3 wgrad 2160x2160x11968", 2160, 2160, 11968), ("c3 wgrad 1520x1520x11968", 1520, 1520, 11968),
          ("c3 fwd 11968x2160x2160", 11968, 2160, 2160), ("c3 fwd 11968x1200x1200", 11968, 1200, 1200), ("c3 out 11968x9998x2160", 11968, 9998, 2160),
          ("c5 wgrad 1024x4096x3008", 1024, 4096, 3008), ("c5 fwd 3008x4096x1024", 3008, 4096, 1024), ("c5 wgrad 1024x1024x3008", 1024, 1024, 3008)]
for name, M, N, K in shapes:
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Ab, _ = ops.bf16_convert(A)
    Bb, _ = ops.bf16_convert(B)
    M8, N8 = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    At = torch.zeros(K, M8, dtype=torch.bfloat16, device="cuda"); At[:, :M] = Ab[:, :K].T
    Bt = torch.zeros(K, N8, dtype=torch.bfloat16, device="cuda"); Bt[:, :N] = Bb[:, :K].T
    fl = 2.0 * M * N * K
    base = ops.gemm_bf16(Ab, Bb, K)
    out = [f"{name:30s} k-contiguous {timeit(lambda: ops.gemm_bf16(Ab, Bb, K)):7.1f} us"]
    for ta, tb in ((True, True), (True, False), (False, True)):
        fn = lambda: ops.gemm_bf16_ex(At if ta else Ab, Bt if tb else Bb, M, N, K, a_kmajor=ta, b_kmajor=tb)
        ok = torch.equal(fn(), base)
        us = timeit(fn)
        out.append(f"ta={int(ta)} tb={int(tb)} {us:7.1f} us ({fl / us / 1e6:6.0f} TF/s){'' if ok else ' MISMATCH'}")
    print(" | ".join(out), flush=True)
