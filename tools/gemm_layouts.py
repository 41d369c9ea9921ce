"""GEMM throughput by operand layout (no epilogue extras): isolates the k-rows / k-contiguous staging paths"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (M, N, K) in [(24000, 2400, 800), (24000, 800, 2400), (12000, 3360, 1120), (6016, 4320, 1440), (4096, 4096, 4096), (24064, 2432, 800)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda") / K ** 0.5
    At, Bt = A.t().contiguous(), B.t().contiguous()
    out = []
    for akc in (True, False):
        for bkc in (True, False):
            a = A if akc else At
            b = Bt if bkc else B
            t = timeit(lambda: ops.gemm(a, b, akc, bkc))
            out.append(f"a{'K' if akc else 'r'}b{'K' if bkc else 'r'} {2.0 * M * N * K / t / 1e9:6.1f}")
    print(f"[layouts] M={M} N={N} K={K}: " + " | ".join(out) + "  TF/s", flush=True)
