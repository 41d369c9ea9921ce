"""producer-wave experiment kernel (gemm_p5.hpp, probe library, W2L_GEMM_P5=1) against the shipped 128 x 128 kernel: correctness
(max relative difference of the results) and time per shape.   python tools/gemm_p5.py  (spawns itself once per variant)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p5 in ("0", "1", "0", "1"):
        env = dict(os.environ, W2L_HIP_SO=os.path.join(root, "wav2letter_amd", "libw2l_hip_probe.so"), W2L_GEMM_P5=p5, W2L_GEMM_T160="0", W2L_GEMM_SK="0")
        subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
import torch
from wav2letter_amd import _lib
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tag = os.environ["W2L_GEMM_P5"]
for name, M, K, N in [("fc1 fwd", 24000, 800, 2400), ("fc3 fwd", 6016, 1440, 4320), ("4096^3", 4096, 4096, 4096), ("small", 1000, 96, 520)]:
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5; b = torch.randn(N, device="cuda", generator=g)
    wt = w.t().contiguous(); y = torch.empty(M, N, device="cuda"); y2 = torch.empty(M, N, device="cuda")
    fwd = lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)              # A k-contiguous, B k-rows
    kck = lambda: L.w2l_linear_backward_data(M, N, K, x.data_ptr(), wt.data_ptr(), y2.data_ptr(), 0, None, 1.0, s)         # both k-contiguous: y2 = x wt^T
    fwd(); kck(); torch.cuda.synchronize()
    ref = torch.relu(x.double() @ w.double() + b.double()) if M * N * K < 3e11 else None
    e1 = float((y.double() - ref).abs().max() / ref.abs().max()) if ref is not None else -1
    ref2 = x.double() @ w.double() if ref is not None else None
    e2 = float((y2.double() - ref2).abs().max() / ref2.abs().max()) if ref is not None else -1
    t1, t2 = timeit(fwd), timeit(kck)
    fl = 2.0 * M * N * K
    print(f"[p5={tag}] {name:8s} M={M} K={K} N={N}: A kc / B rows {t1:.0f} us {fl / t1 / 1e6:.1f} TF (err {e1:.1e}) | both kc {t2:.0f} us {fl / t2 / 1e6:.1f} TF (err {e2:.1e})", flush=True)
