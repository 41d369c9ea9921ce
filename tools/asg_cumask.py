"""ASG (FCC + FAC) at the conv_glu criterion shape: does pinning the two serial scans to disjoint halves of the chip
(hipExtStreamCreateWithCUMask) remove their interference?   python tools/asg_cumask.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ASGLoss, CriterionScaleMode
import wav2letter_amd.criterion as crit_mod

hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi, ncu=256):
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(lo, hi):
        mask[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    st = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
    assert st == 0, st
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    device = "cuda"
    B, T, N, L = 64, 2000, 30, 300
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, T, N, generator=g).to(device).requires_grad_(True)
    tgt = torch.full((B, L), -1, dtype=torch.int32)
    for b in range(B):
        l = int(torch.randint(60, L + 1, (1,), generator=g))
        y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
        for i in range(1, l):
            if y[i] == y[i - 1]:
                y[i] = (y[i] + 1) % 28
        tgt[b, :l] = y
    tgt = tgt.to(device)
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).to(device)
    print(f"as is:            fwd {timeit(lambda: crit(x, tgt)):.4f} ms   fwd+bwd {timeit(lambda: crit(x, tgt).sum().backward()):.4f} ms")
    print(f"alone:            fcc {timeit(lambda: crit.fcc(x, tgt)):.4f}  fac {timeit(lambda: crit.fac(x, tgt)):.4f} ms")
    # FAC pinned to the upper half of the chip
    crit._side = masked_stream(128, 256)
    print(f"FAC on CU 128-255: fwd {timeit(lambda: crit(x, tgt)):.4f} ms   fwd+bwd {timeit(lambda: crit(x, tgt).sum().backward()):.4f} ms")
    # both pinned: run the whole criterion on a stream masked to the lower half, FAC on the upper half
    lower = masked_stream(0, 128)

    def both():
        cur = torch.cuda.current_stream()
        lower.wait_stream(cur)
        with torch.cuda.stream(lower):
            out = crit(x, tgt)
        cur.wait_stream(lower)
        return out
    print(f"FCC on CU 0-127, FAC on 128-255: fwd {timeit(both):.4f} ms   fwd+bwd {timeit(lambda: both().sum().backward()):.4f} ms")
    for lo, hi in ((0, 64), (0, 32)):
        crit._side = masked_stream(128, 128 + (hi - lo) * 2 if False else 256)
        lower2 = masked_stream(lo, hi)

        def both2():
            cur = torch.cuda.current_stream()
            lower2.wait_stream(cur)
            with torch.cuda.stream(lower2):
                out = crit(x, tgt)
            cur.wait_stream(lower2)
            return out
        print(f"FCC on CU {lo}-{hi - 1}: fwd {timeit(both2):.4f} ms")


if __name__ == "__main__":
    main()


def cpu_bound_check():
    """is the criterion call host-bound?  wall time per call without any device synchronisation between calls vs the
    device time of the same calls"""
    import time
    device = "cuda"
    B, T, N, L = 64, 2000, 30, 300
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, T, N, generator=g).to(device).requires_grad_(True)
    tgt = torch.randint(0, 28, (B, L), generator=g, dtype=torch.int32)
    tgt[:, 1::2] = (tgt[:, 1::2] + 1 + tgt[:, 0::2]) % 28
    tgt = tgt.to(device)
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).to(device)
    for _ in range(5):
        crit(x, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        crit(x, tgt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host time per forward call (enqueue only) {(t1 - t0) / 50 * 1e3:.3f} ms; until the device drained {(t2 - t0) / 50 * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(50):
        crit(x, tgt).sum().backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host time per forward+backward call {(t1 - t0) / 50 * 1e3:.3f} ms; until the device drained {(t2 - t0) / 50 * 1e3:.3f} ms")


if __name__ == "__main__":
    cpu_bound_check()
