"""replay one case of tools/exp/criterion_fuzz.py (seed, index) for CTC and print device / oracle values"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
seed, want_c = int(sys.argv[1]), int(sys.argv[2])
xs_c = (0.1, 1.0, 5.0, 20.0); as_c = (0.0, 0.3, 2.0, 8.0)
rng = np.random.default_rng(seed)
for c in range(want_c + 1):
    N = int(rng.choice([3, 5, 16, 29, 30, 31, 32, 40, 64, 65, 100]))
    T = int(rng.choice([1, 2, 17, 50, 300, 1000, 2000]))
    B = int(rng.integers(1, 4))
    Lmax = int(rng.choice([1, 2, 7, 64, 65, 128, 200, 300]))
    xs = float(rng.choice(xs_c)); as_ = float(rng.choice(as_c))
    diag = float(rng.choice([0.0, 4.0])); mode = int(rng.choice([0, 1, 2, 3, 4]))
    x = (rng.normal(size=(B, T, N)) * xs).astype(np.float32)
    A = (rng.normal(size=(N, N)) * as_ + np.eye(N) * diag).astype(np.float32)
    tgt = np.full((B, Lmax), -1, np.int32)
    for b in range(B):
        l = int(rng.integers(1, min(Lmax, T) + 1))
        tgt[b, :l] = rng.integers(0, N, size=l)
    w = rng.uniform(0.5, 1.5, size=B)
tg = np.where(tgt >= 0, np.minimum(tgt, N - 2), -1).astype(np.int32)
print("B T N Lmax mode", B, T, N, Lmax, mode, "target", tg[:, :4], "w", w)
print("x", x)
from oracle import pyoracle as O
o = O.CTC(x, tg, scale_mode=mode)
ol = o.forward(); og = o.backward(w)
print("oracle loss", ol, "grad", og)
import torch
if torch.cuda.is_available():
    from wav2letter_amd import CTCLoss
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    loss = CTCLoss(mode)(xt, torch.from_numpy(tg).cuda())
    (loss * torch.from_numpy(w.astype(np.float32)).cuda()).sum().backward()
    print("device loss", loss.detach().cpu().numpy(), "grad", xt.grad.cpu().numpy())
