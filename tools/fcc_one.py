import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd.criterion import CriterionScaleMode, FullConnectionCriterion
B, T, N = 32, 40, 9998
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B, T, N, generator=g).cuda()
tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
crit = FullConnectionCriterion(N, CriterionScaleMode.NONE).cuda()
crit.transitions.data = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
for _ in range(2):
    crit(x, tgt)
torch.cuda.synchronize()
