"""where the Python autograd wrapper of ASGLoss spends its time (one-call path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wav2letter_amd import ASGLoss, CriterionScaleMode
dev = torch.device("cuda:0")
B, T, N, L = 64, 2000, 30, 300
g = torch.Generator(device="cpu").manual_seed(4)
x = torch.randn(B, T, N, generator=g).to(dev).requires_grad_(True)
tgt = torch.full((B, L), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(60, L + 1, (1,), generator=g))
    tgt[b, :l] = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
tgt = tgt.to(dev)
crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).to(dev)
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4), round((t1 - t0) / n * 1e3, 4)
print("fwd", timeit(lambda: crit(x, tgt)))
print("fwd+backward()", timeit(lambda: crit(x, tgt).sum().backward()))
print("fwd+autograd.grad", timeit(lambda: torch.autograd.grad(crit(x, tgt).sum(), [x, crit.transitions])))
xd = x.detach().requires_grad_(True)
print("fwd+grad wrt x only", timeit(lambda: torch.autograd.grad(crit(xd, tgt).sum(), [xd])))
w = torch.ones(B, device=dev)
print("fwd+backward(ones)", timeit(lambda: crit(x, tgt).backward(w)))
