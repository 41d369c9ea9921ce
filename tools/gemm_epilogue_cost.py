"""what the epilogue operands cost: plain product vs + addend vs + dropout vs + both, on the lin2 / dX shapes of the headline and
of config 3.   python tools/gemm_epilogue_cost.py"""
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

for M, K, N in [(24000, 2400, 800), (12000, 3360, 1120), (6016, 4320, 1440), (11968, 3600, 1200), (11968, 6480, 2160)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    add = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); addx = torch.randn(M, K, device="cuda")
    for rep in range(2):
        t0 = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, s))
        t1 = timeit(lambda: L.w2l_linear_forward_dropout_add(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), add.data_ptr(), y.data_ptr(), 0, 0.0, 1, 2, s))
        t2 = timeit(lambda: L.w2l_linear_forward_dropout(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 0.1, 1, 2, s))
        t3 = timeit(lambda: L.w2l_linear_forward_dropout_add(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), add.data_ptr(), y.data_ptr(), 0, 0.1, 1, 2, s))
        # dX of the mirrored layer (in = N, out = K): [M][N] = dy[M][K] w'[N][K]^T
        t4 = timeit(lambda: L.w2l_linear_backward_data(M, N, K, x.data_ptr(), w.data_ptr(), y.data_ptr(), 0, None, 1.0, s))
        t5 = timeit(lambda: L.w2l_linear_backward_data_add(M, N, K, x.data_ptr(), w.data_ptr(), add.data_ptr(), y.data_ptr(), s))
        print(f"M={M} K={K} N={N}: fwd {t0:.0f} us | +addend {t1:.0f} | +dropout {t2:.0f} | +both {t3:.0f} || dX {t4:.0f} | dX+addend {t5:.0f}", flush=True)
