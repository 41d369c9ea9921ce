"""final fl::Linear of the TDS-CTC recipe (N = 9998) and the dA GEMM of the ASG stress shape: relaxed-alignment
LDS-DMA path (default) against the register-staged kernel (W2L_GEMM_UNALIGNED=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
a = torch.randn(4096, 4096, device="cuda"); c = torch.empty(4096, 4096, device="cuda")
timeit(lambda: L.w2l_linear_forward(4096, 4096, 4096, a.data_ptr(), a.data_ptr(), None, c.data_ptr(), 0, s), n=60)
M, K, N = 6016, 1440, 9998
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N, device="cuda")
fl = 2.0 * M * N * K
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["W2L_GEMM_UNALIGNED"] = mode
    tf = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, s))
    yv = y.clone()
    td = timeit(lambda: L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s))
    dxv = dx.clone()
    tw = timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))
    res[mode] = (yv, dxv, dw.clone())
    print(f"[final unaligned={mode}] M={M} K={K} N={N}: fwd {tf * 1e3:.0f} us {fl / tf / 1e9:.1f} TF | dX {td * 1e3:.0f} us {fl / td / 1e9:.1f} TF | "
          f"dW {tw * 1e3:.0f} us {fl / tw / 1e9:.1f} TF", flush=True)
for i, nm in enumerate(("y", "dx", "dw")):
    d = (res["0"][i] - res["1"][i]).abs().max().item() / res["0"][i].abs().max().item()
    print(f"[final] {nm}: max rel diff between the two paths {d:.2e}", flush=True)
