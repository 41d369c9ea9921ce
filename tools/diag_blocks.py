"""Localise the gradient error of the role-swapped TDS conv inside a network: reduced archs with k TDS blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from oracle import pyoracle as O
from wav2letter_amd import _lib
from test_gpu_trainer import build, rel


def run(tag, c, nblocks, T, B=2, l2=0, stages=None):
    rng = np.random.default_rng(21)
    nfeat, nlabel, L = 80, 40, 5
    lines = ["V -1 NFEAT 1 0"]
    cin = 1
    for cc, nb, ll in (stages or [(c, nblocks, l2)]):
        lines += [f"C2 {cin} {cc} 21 1 2 1 -1 -1", "R", "DO 0.0", "LN 0 1 2"] + [f"TDS {cc} 21 80 0.0 {ll}"] * nb
        cin = cc
    c = cin
    lines += [f"V 0 {c * 80} 1 0", "RO 1 0 3 2", f"L {c * 80} NLABEL"]
    arch = "\n".join(lines) + "\n"
    tr, ref, params, _ = build(arch, nfeat, nlabel, "ctc", 4, 0.0, rng, B, T, L)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    tgt = rng.integers(0, nlabel - 1, size=(B, L)).astype(np.int32)
    xd = torch.tensor(x.reshape(B, nfeat, T)).cuda()
    td = torch.tensor(tgt).cuda()
    em_ref = ref.forward(x, params)
    em = tr.forward(xd, train=False).cpu().numpy()
    loss = tr.forward_backward(xd, td).cpu().numpy()
    o = O.CTC(em_ref, tgt, scale_mode=4)
    o.forward()
    grads = ref.backward(o.backward().astype(np.float32), len(params))
    g = tr.grads.cpu().numpy()
    table = tr.param_table()
    errs = [rel(tr.export_from(i, g), want) for i, want in enumerate(grads)]
    bad = [(i, table[i][0], float(f"{e:.0e}")) for i, e in enumerate(errs) if e > 2e-5]
    print(f"[{tag}] stages={stages or [(c, nblocks, l2)]} T={T}: emission err {rel(em, em_ref):.1e}; worst grad err {max(errs):.1e}; "
          f"params over 2e-5: {len(bad)}/{len(errs)} first {bad[:4]} last {bad[-3:]}")


for kw in [dict(c=10, nblocks=2, T=96, l2=2400), dict(c=10, nblocks=1, T=96, stages=[(10, 1, 0), (14, 1, 0)]),
           dict(c=10, nblocks=1, T=96, stages=[(10, 1, 0), (14, 1, 0), (18, 1, 0)]),
           dict(c=10, nblocks=1, T=96, stages=[(10, 5, 2400), (14, 6, 3360), (18, 10, 4320)])]:
    run("product", **kw)
    with _lib.use_probe():
        os.environ["W2L_TDS_RS_OFF"] = "1"
        run("probe RS off", **kw)
        os.environ.pop("W2L_TDS_RS_OFF")
