"""dA GEMM of the ASG stress shape: dA[9998][9998] = G^T E, both operands k-rows with ld = 9998, K = (T-1) B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops

def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (M, N, K) in [(9998, 9998, 47968), (9998, 9998, 4800), (10000, 10000, 4800), (9998, 10000, 4800)]:
    At = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda")
    for mode in ("0", "1", "0", "1"):
        os.environ["W2L_GEMM_UNALIGNED"] = mode
        t = timeit(lambda: ops.gemm(At, B, False, False))
        print(f"[da unaligned={mode}] M={M} N={N} K={K}: {t:.2f} ms {2.0 * M * N * K / t / 1e9:.1f} TF", flush=True)
    del At, B
