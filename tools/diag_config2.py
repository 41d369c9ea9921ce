"""Diagnostic for tests/test_gpu_trainer.py::test_tds_ctc_config2_full_network_end_to_end: per-parameter gradient error of
the product library and of the previous TDS conv kernels (probe library, W2L_TDS_RS_OFF=1), and how many ReLU masks of
the TDS convolutions differ from the reference's (a pre-activation within rounding of zero flips the mask: relu'(0+-))."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from oracle import pyoracle as O
from wav2letter_amd import _lib, recipes
from test_gpu_trainer import build, rel


def run(tag):
    rng = np.random.default_rng(21)
    nfeat, nlabel, B, T, L = 80, 9998, 2, 96, 5
    arch = re.sub(r"(TDS \d+ \d+ \d+) [0-9.]+", r"\1 0.0", recipes.tds_ctc_arch())
    arch = "\n".join(l for l in arch.splitlines() if not l.startswith("SAUG")) + "\n"
    arch = re.sub(r"^DO [0-9.]+$", "DO 0.0", arch, flags=re.M)
    tr, ref, params, _ = build(arch, nfeat, nlabel, "ctc", 4, 0.0, rng, B, T, L)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    tgt = np.full((B, L), -1, np.int32)
    tgt[0, :5] = [17, 4021, 9996, 3, 3]
    tgt[1, :2] = [9000, 12]
    xd = torch.tensor(x.reshape(B, nfeat, T)).cuda()
    td = torch.tensor(tgt).cuda()
    em_ref = ref.forward(x, params)
    loss = tr.forward_backward(xd, td).cpu().numpy()
    o = O.CTC(em_ref, tgt, scale_mode=4)
    o.forward()
    grads = ref.backward(o.backward().astype(np.float32), len(params))
    g = tr.grads.cpu().numpy()
    table = tr.param_table()
    errs = [(rel(tr.export_from(i, g), want), i, table[i][0]) for i, want in enumerate(grads)]
    bad = [e for e in errs if e[0] >= 2e-4]
    print(f"[{tag}] params over 2e-4: {len(bad)} of {len(errs)}; worst: {sorted(errs, reverse=True)[:6]}")
    print(f"[{tag}] first 12: {[(i, n, float(f'{e:.2e}')) for e, i, n in errs[:12]]}")
    # smallest |pre-activation| of the TDS convolutions in the reference (where a ReLU mask can flip)
    mins = []
    for rec in ref.tape:
        if rec[0] == "TDS":
            a = rec[2]["a"]
            mins.append(float(np.abs(a).min()))
    print(f"[{tag}] smallest |conv pre-activation| per TDS block (reference): {['%.1e' % m for m in mins]}")


run("product (role-swapped conv)")
with _lib.use_probe():
    run("probe, role-swapped conv")
    os.environ["W2L_TDS_RS_OFF"] = "1"
    run("probe, previous conv kernels")
    os.environ.pop("W2L_TDS_RS_OFF")
    os.environ["W2L_TDS_RSF_OFF"] = "1"
    run("probe, role-swapped fwd/bwd-data, previous filter")
