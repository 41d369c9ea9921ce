"""The alpha / beta passes of FullConnectionCriterion at the north-star stress shape (B = 32, N = 9998, T frames), with and
WITHOUT the bench's event brackets around every stream launch; repeated runs must be bit-identical.  (profiles/r05_run24_*: the
run over the step-kernel variants W2L_FCC_STEPV -- unconditional slab loads / c_t from cfin / nontemporal stores -- that are
the only form since; they were worth 0.3 us per frame, the event brackets 4.6.)   python tools/fcc_step_variants.py [T] [repeats]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
from wav2letter_amd.criterion import CriterionScaleMode, FullConnectionCriterion

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
vals = list(range(int(sys.argv[2]) if len(sys.argv) > 2 else 3))
B, N = 32, 9998
g = torch.Generator(device="cpu").manual_seed(7)
x = torch.randn(B, T, N, generator=g).cuda().requires_grad_(True)
tgt = torch.zeros(B, 8, dtype=torch.int32).cuda()
A = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
step_bytes = 4.0 * N * N + 8.0 * B * N


def kind(L, k):
    n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
    L.w2l_profile_report_kind(k, C.byref(n_), C.byref(ms_), C.byref(w_))
    return n_.value, ms_.value, w_.value


def run(L, events):
    crit = FullConnectionCriterion(N, CriterionScaleMode.TARGET_SZ_SQRT).cuda()
    crit.transitions.data = A
    loss = crit(x, tgt)              # warm-up (workspace, gradient buffers)
    loss.sum().backward()
    x.grad = None
    crit.transitions.grad = None
    torch.cuda.synchronize()
    if events:
        L.w2l_profile_enable(1)
    t0 = time.perf_counter()
    loss = crit(x, tgt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ka = kind(L, 3) if events else None
    if events:
        L.w2l_profile_enable(1)
    loss.sum().backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    kb = kind(L, 3) if events else None
    kd = kind(L, 0) if events else None
    if events:
        L.w2l_profile_enable(0)
    sig = (loss.detach().clone(), x.grad.double().sum().item(), x.grad.abs().double().sum().item(),
           crit.transitions.grad.double().sum().item(), crit.transitions.grad.abs().double().sum().item())
    x.grad = None
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, ka, kb, kd, sig


ref = None
dA_ms = None
for v in vals:
    with _lib.use_probe():
        L = _lib.lib()
        if dA_ms is None:
            f_, b_, ka, kb, kd, _ = run(L, True)
            dA_ms = kd[1]
            print(f"with event brackets: forward {f_:.2f} ms = {f_ * 1e3 / (T - 1):.2f} us per frame, backward {b_:.2f} ms, dA GEMM {dA_ms:.2f} ms, "
                  f"beta recursion {b_ - dA_ms:.2f} ms = {(b_ - dA_ms) * 1e3 / (T - 1):.2f} us per frame; stream kernel alone alpha {ka[1] * 1e3 / ka[0]:.2f} us, "
                  f"beta {kb[1] * 1e3 / kb[0]:.2f} us", flush=True)
        f, b, _, _, _, sig = run(L, False)
    if ref is None:
        ref = sig
    same = torch.equal(sig[0], ref[0]) and sig[1:] == ref[1:]
    beta = b - dA_ms
    print(f"run {v}: alpha pass {f:8.2f} ms = {f * 1e3 / (T - 1):6.2f} us per frame = {step_bytes * (T - 1) / f / 8e9:.4f} of 8 TB/s; "
          f"backward {b:8.2f} ms, beta recursion {beta:8.2f} ms = {beta * 1e3 / (T - 1):6.2f} us per frame = {step_bytes * (T - 1) / beta / 8e9:.4f}; "
          f"identical to the first run: {same}", flush=True)
