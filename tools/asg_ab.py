"""ASG criterion at the bench shape (B = 64, T = 2000, N = 30, L <= 300): forward + backward + Viterbi, n times -- run under
rocprofv3 --kernel-trace (tools/prof.sh) for per-kernel times.  argv[1] = "new" | "old" (old: probe library, W2L_ASG_OLD=1 must
be in the environment), argv[2] = repetitions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wav2letter_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "old":
    _lib.use_probe().__enter__()
from wav2letter_amd import ASGLoss, CriterionScaleMode  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
for _ in range(n):
    crit(x, tgt).sum().backward()
    crit.viterbiPath(x.detach())
torch.cuda.synchronize()
# serial (one criterion at a time, no side stream): the kernels' own durations without co-running neighbours
for _ in range(n):
    crit.fcc(x, tgt).sum().backward()
    torch.cuda.synchronize()
    crit.fac(x, tgt).sum().backward()
    torch.cuda.synchronize()
print("done")
