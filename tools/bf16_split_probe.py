"""PROBE (not a product path): fp32 = bf16 hi + bf16 lo operand split on the bf16 MFMA -- a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi
(the lo.lo term is 2^-16 of the product) -- against the fp32 MFMA GEMM the product uses and the float64 product, on TDS
fc shapes.  Reports the error next to the 1e-4 parity bar and the time of the three bf16 launches + the four extra image
conversions against one fp32 launch.  usage: bf16_split_probe.py [M N K]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops

def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())

def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

shapes = [(6016, 1200, 1200), (6016, 2160, 2160), (24000, 800, 800), (24000, 1440, 1440)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
for M, N, K in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    want = x.double() @ w.double().T
    wt = w.t().contiguous()
    res = {}
    def f32():
        res["y"] = ops.linear_forward(x, wt, None)
    out = torch.empty(M, N, device="cuda")
    def split():
        xh, _ = ops.bf16_convert(x); wh, _ = ops.bf16_convert(w)
        xl, _ = ops.bf16_convert(x - xh[:, :K].float()); wl, _ = ops.bf16_convert(w - wh[:, :K].float())
        ops.gemm_bf16(xh, wh, K, out=out)
        ops.gemm_bf16(xh, wl, K, out=out, accumulate=True)
        ops.gemm_bf16(xl, wh, K, out=out, accumulate=True)
    def single():
        xh, _ = ops.bf16_convert(x); wh, _ = ops.bf16_convert(w)
        ops.gemm_bf16(xh, wh, K, out=out)
    t32, t3, t1 = timed(f32), timed(split), timed(single)
    f32(); e32 = rel(res["y"], want)
    split(); e3 = rel(out, want)
    single(); e1 = rel(out, want)
    print("split probe M=%d N=%d K=%d: max-rel error vs float64: fp32 MFMA %.2e | bf16 hi/lo 3-term %.2e | plain bf16 %.2e (bar 1e-4);"
          " time us: fp32 %.1f | split (3 GEMMs + 4 conversions + 2 subtractions) %.1f | plain bf16 (incl. 2 conversions) %.1f"
          % (M, N, K, e32, e3, e1, t32, t3, t1), flush=True)
