"""ASG criteria (FCC, FAC) at the conv_glu criterion shape with transition rows tens to hundreds of nats wide -- the inputs that send every
utterance to the log-domain fallback kernels -- against the fp64 oracle.   python tools/exp/asg_wide_transitions.py [sigma ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import pyoracle as O
from wav2letter_amd import ForceAlignmentCriterion, FullConnectionCriterion

sig = [float(v) for v in sys.argv[1:]] or [8.0, 12.0, 25.0, 50.0]
B, T, N, L = 4, 2000, 30, 300
rng = np.random.default_rng(3)
x = rng.normal(size=(B, T, N)).astype(np.float32)
tgt = np.full((B, L), -1, np.int32)
for b in range(B):
    l = int(rng.integers(60, L + 1))
    y = rng.integers(0, 28, size=l)
    for i in range(1, l):
        if y[i] == y[i - 1]:
            y[i] = (y[i] + 1) % 28
    tgt[b, :l] = y
ts = O.batch_target_size(tgt, T)
for s in sig:
    A = (np.eye(N) * 4 + rng.normal(size=(N, N)) * s).astype(np.float32)
    for name, cls, orc in (("FCC", FullConnectionCriterion, lambda: O.FCC(x, A, ts, 4)), ("FAC", ForceAlignmentCriterion, lambda: O.FAC(x, A, tgt, scale_mode=4))):
        crit = cls(N, 4).cuda()
        crit.transitions.data = torch.from_numpy(A).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        loss = crit(xt, torch.from_numpy(tgt).cuda())
        loss.sum().backward()
        o = orc()
        ol = o.forward()
        odx, odA = o.backward(np.ones(B))
        got = loss.detach().cpu().numpy()
        gdx = xt.grad.cpu().numpy(); gdA = crit.transitions.grad.cpu().numpy()
        rel = np.abs(got - ol).max() / max(1.0, np.abs(ol).max())
        print(f"sigma {s:5.1f} {name}: loss finite {np.isfinite(got).all()} (oracle finite {np.isfinite(ol).all()}) rel err {rel:.2e}; dx finite {np.isfinite(gdx).all()} "
              f"err {np.abs(gdx - odx).max() / max(1e-30, np.abs(odx).max()):.2e}; dA finite {np.isfinite(gdA).all()} err {np.abs(gdA - odA).max() / max(1e-30, np.abs(odA).max()):.2e}; "
              f"loss[0] {got[0]:.6g} oracle {ol[0]:.6g}", flush=True)
