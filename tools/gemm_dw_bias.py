"""weight gradient alone (+ the separate column-sum launch) against w2l_linear_backward_weight_bias (column sums riding on the
product) on the fl::Linear shapes of the headline step.   python tools/gemm_dw_bias.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, K, N in [(24000, 800, 2400), (24000, 2400, 800), (12000, 1120, 3360), (12000, 3360, 1120), (6016, 1440, 4320), (6016, 4320, 1440)]:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda"); dw = torch.empty(K, N, device="cuda"); db = torch.empty(N, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))
        tc = timeit(lambda: L.w2l_colsum(dy.data_ptr(), db.data_ptr(), M, N, s))
        t1 = timeit(lambda: L.w2l_linear_backward_weight_bias(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s))
        print(f"M={M} in={K} out={N}: dW {t0:.1f} us + colsum {tc:.1f} us = {t0 + tc:.1f} | fused {t1:.1f} us ({t1 - t0:+.1f} on the product)", flush=True)
