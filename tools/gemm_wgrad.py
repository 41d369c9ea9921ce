"""weight-gradient GEMMs of the TDS-CTC step (M x N small, K = B*T huge, both operands k-major): aligned K split (probe library,
W2L_GEMM_KSPLIT=1) against the product's stream-K ranges; max |difference| against a float64 product on a slice"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops, _lib


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


a0 = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    a0 @ a0
for (M, N, K) in [(800, 2400, 24000), (2400, 800, 24000), (1120, 3360, 12000), (3360, 1120, 12000), (1440, 4320, 6016), (4320, 1440, 6016),
                  (1440, 9998, 6016)]:
    At = torch.randn(K, M, device="cuda")            # [K][M]: A k-major
    B = torch.randn(K, N, device="cuda") / K ** 0.5  # [K][N]
    os.environ["W2L_GEMM_KSPLIT"] = "1"
    with _lib.use_probe():
        got = ops.gemm(At, B, False, False)
    os.environ.pop("W2L_GEMM_KSPLIT")
    ref = (At[:, :64].double().t() @ B.double())
    err = ((got[:64].double() - ref).abs().max() / ref.abs().max()).item()
    old = ops.gemm(At, B, False, False)
    tn, to = [], []
    for rep in range(4):   # interleaved A/B: the two variants alternate on the same box, minimum of four
        os.environ["W2L_GEMM_KSPLIT"] = "1"
        with _lib.use_probe():
            tn.append(timeit(lambda: ops.gemm(At, B, False, False), n=10, warm=2))
        os.environ.pop("W2L_GEMM_KSPLIT")
        to.append(timeit(lambda: ops.gemm(At, B, False, False), n=10, warm=2))
    t_new, t_old = min(tn), min(to)
    d = ((got.double() - old.double()).abs().max() / old.double().abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"[wgrad] M={M} N={N} K={K}: aligned K split {t_new * 1e3:7.1f} us = {fl / t_new / 1e9:6.1f} TF/s | stream-K ranges {t_old * 1e3:7.1f} us = "
          f"{fl / t_old / 1e9:6.1f} TF/s | vs fp64 {err:.1e}  vs stream-K {d:.1e}", flush=True)
