"""ASG forward (+ backward) at the conv_glu criterion shape, a few calls: run under `rocprofv3 --kernel-trace` for the timeline"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ASGLoss, CriterionScaleMode
B, T, N, L = 64, 2000, 30, 300
g = torch.Generator(device="cpu").manual_seed(4)
x = torch.randn(B, T, N, generator=g).cuda().requires_grad_(True)
tgt = torch.full((B, L), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(60, L + 1, (1,), generator=g))
    y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
    for i in range(1, l):
        if y[i] == y[i - 1]:
            y[i] = (y[i] + 1) % 28
    tgt[b, :l] = y
tgt = tgt.cuda()
crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).cuda()
for _ in range(3):
    crit(x, tgt).sum().backward()
torch.cuda.synchronize()
