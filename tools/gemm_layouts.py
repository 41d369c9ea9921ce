"""GEMM throughput by operand layout (no epilogue extras): isolates the k-rows / k-contiguous staging paths.
  python tools/gemm_layouts.py                     all four layouts on the step's shapes
  python tools/gemm_layouts.py --ak [MxNxK ...]    aKbK against aKbr over (M, K) variations: which dimension triggers the
                                                   both-K-contiguous slowdown (was tools/gemm_layouts2.py)"""
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


if "--ak" in sys.argv:
    shapes = [(24000, 2400, 1120), (12000, 2400, 800), (6000, 2400, 800), (48000, 2400, 800), (24000, 2400, 768),
              (24000, 2400, 832), (24000, 2400, 1024), (24000, 2400, 864), (24000, 2432, 1600), (24000, 1280, 800), (24000, 4864, 800)]
    given = [a for a in sys.argv[1:] if a != "--ak"]
    if given:
        shapes = [tuple(int(v) for v in s.split("x")) for s in given]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda") / K ** 0.5
        Bt = B.t().contiguous()
        t1 = timeit(lambda: ops.gemm(A, Bt, True, True))
        t2 = timeit(lambda: ops.gemm(A, B, True, False))
        f = 2.0 * M * N * K / 1e9
        print(f"[layouts2] M={M} N={N} K={K}: aKbK {f / t1:6.1f} ({t1 * 1e3:.0f} us) | aKbr {f / t2:6.1f} ({t2 * 1e3:.0f} us)  TF/s", flush=True)
    sys.exit(0)

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
