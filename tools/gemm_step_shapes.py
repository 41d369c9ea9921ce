"""The fp32 GEMM launches of the headline step (profiles/r06_run13_gemm_per_shape.md), timed stand-alone per role, with the
step-weighted total: a proxy of `gemm_ms_per_step` for A/B runs of kernel variants (W2L_HIP_SO=.../libw2l_hip_probe.so + a
W2L_GEMM_* switch; the switches are read once per process).   python tools/gemm_step_shapes.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib

tag = sys.argv[1] if len(sys.argv) > 1 else ""
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# (rows M, in, out, Linear layers of that shape per step)
layers = [(24000, 800, 2400, 5), (24000, 2400, 800, 5), (12000, 1120, 3360, 6), (12000, 3360, 1120, 6),
          (6016, 1440, 4320, 10), (6016, 4320, 1440, 10), (6016, 1440, 9998, 1)]
if os.environ.get("SHAPES") == "short":
    layers = [(24000, 800, 2400, 5), (12000, 1120, 3360, 6), (6016, 1440, 4320, 10), (6016, 4320, 1440, 10)]
tot = 0.0; flops = 0.0
for M, K, N, cnt in layers:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N, device="cuda")
    tf = timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s))
    td = timeit(lambda: L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s))
    tw = timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))
    fl = 2.0 * M * N * K
    tot += cnt * (tf + td + tw); flops += 3 * cnt * fl
    print(f"[{tag}] M={M} in={K} out={N}: fwd {tf * 1e3:.0f} us {fl / tf / 1e9:.1f} TF | dX {td * 1e3:.0f} us {fl / td / 1e9:.1f} TF | "
          f"dW {tw * 1e3:.0f} us {fl / tw / 1e9:.1f} TF", flush=True)
    del x, w, b, dy, y, dx, dw
print(f"[{tag}] step-weighted: {tot:.2f} ms, {flops / tot / 1e9:.1f} TFLOP/s = {flops / tot / 1e9 / 157.3:.3f} of peak", flush=True)
