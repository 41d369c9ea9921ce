"""do the LDS-DMA kernels work on rows that are only dword aligned (odd leading dimensions)?  W2L_GEMM_UNALIGNED=2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(300, 999, 64), (999, 262, 96), (257, 321, 353 * 3 // 32 * 32), (4096, 1001, 1024)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda") / K ** 0.5
    want = (A.double() @ B.double())
    for akc in (True, False):
        for bkc in (True, False):
            a = A if akc else A.t().contiguous()
            b = B.t().contiguous() if bkc else B
            got = ops.gemm(a, b, akc, bkc)
            err = ((got.double() - want).abs().max() / want.abs().max()).item()
            print(f"[dword] M={M} N={N} K={K} akc={int(akc)} bkc={int(bkc)}: rel err {err:.2e}", flush=True)
