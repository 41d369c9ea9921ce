"""The ASG alpha recursion at the north-star stress shape (B = 32, N = 9998): what fraction of every worker's share of the
400 MB transition stream to load with the DEFAULT cache policy (so that it can stay in the 256 MiB Infinity Cache between
the T dependent steps) while the rest streams nontemporal (probe library, W2L_FCC_CACHE = per mille).  The policy does not
touch the arithmetic: losses must be bit-identical.   python tools/fcc_cache.py [T] [per-mille values ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
from wav2letter_amd.criterion import CriterionScaleMode, FullConnectionCriterion

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
vals = [int(v) for v in sys.argv[2:]] or [0, 250, 400, 500, 600, 0, 500]
B, N = 32, 9998
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B, T, N, generator=g).cuda().requires_grad_(True)
tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
A = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
step_bytes = 4.0 * N * N + 8.0 * B * N


def run(reps=2, backward=False):
    crit = FullConnectionCriterion(N, CriterionScaleMode.TARGET_SZ_SQRT).cuda()
    crit.transitions.data = A
    loss = crit(x, tgt)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        loss = crit(x, tgt)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    tb = None
    if backward:
        t0 = time.perf_counter()
        loss.sum().backward()
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        x.grad = None
    return loss.detach().clone(), min(ts), tb


ref = None
for v in vals:
    os.environ["W2L_FCC_CACHE"] = str(v)
    with _lib.use_probe():
        loss, dt, tb = run(backward=True)
    if ref is None:
        ref = loss
    print(f"W2L_FCC_CACHE={v:4d}: forward {dt * 1e3:8.2f} ms = {dt * 1e6 / (T - 1):6.2f} us per step = "
          f"{step_bytes * (T - 1) / dt / 1e9:7.1f} GB/s ({step_bytes * (T - 1) / dt / 8e12:.3f} of 8 TB/s); backward {tb * 1e3:8.2f} ms; "
          f"loss identical to the first run: {torch.equal(loss, ref)}", flush=True)
prod, dtp, _ = run()
print(f"product library: forward {dtp * 1e3:8.2f} ms = {dtp * 1e6 / (T - 1):6.2f} us per step; loss identical: {torch.equal(prod, ref)}")
