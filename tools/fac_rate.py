"""FAC forward / backward scan rate (cycles per frame at 2.4 GHz) against the number of waves per utterance"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ASGLoss, CriterionScaleMode
from tools.asg_cumask import timeit
B, T, N = 64, 2000, 30
for L in (60, 120, 180, 300, 500):
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, T, N, generator=g).cuda().requires_grad_(True)
    y = torch.randint(0, 28, (B, L), generator=g, dtype=torch.int32)
    for i in range(1, L):
        same = y[:, i] == y[:, i - 1]
        y[same, i] = (y[same, i] + 1) % 28
    tgt = y.cuda()
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).cuda()
    f = timeit(lambda: crit.fac(x, tgt))
    fb = timeit(lambda: crit.fac(x, tgt).sum().backward())
    print(f"L={L:4d} ({(L + 63) // 64} waves): fac fwd {f:.4f} ms = {f * 1e-3 * 2.4e9 / T:6.0f} cycles/frame   fwd+bwd {fb:.4f} ms  (bwd {(fb - f) * 1e-3 * 2.4e9 / T:6.0f} cycles/frame incl. scatter)")
