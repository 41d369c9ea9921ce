"""Is the strict-gradient failure of a one-block TDS network under the block-Toeplitz kernels a ReLU kink (an input of the block's
ReLU within fp32 rounding of zero that the two summation orders put on different sides) or a kernel error?  Per-parameter
gradient errors of the product library and of the probe library with the previous generation (W2L_TDS_TZ_OFF / TZF_OFF), for a
few seeds; and the smallest |pre-activation| of the TDS block's convolution on the device.   python tools/diag_tz_kink.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from oracle import pyoracle as O
from wav2letter_amd import _lib
from test_gpu_trainer import build, rel


def run(tag, seed, stages, T=96, B=2):
    rng = np.random.default_rng(seed)
    nfeat, nlabel, L = 80, 40, 5
    lines = ["V -1 NFEAT 1 0"]
    cin = 1
    for cc, nb, ll in stages:
        lines += [f"C2 {cin} {cc} 21 1 2 1 -1 -1", "R", "DO 0.0", "LN 0 1 2"] + [f"TDS {cc} 21 80 0.0 {ll}"] * nb
        cin = cc
    lines += [f"V 0 {cin * 80} 1 0", "RO 1 0 3 2", f"L {cin * 80} NLABEL"]
    arch = "\n".join(lines) + "\n"
    tr, ref, params, _ = build(arch, nfeat, nlabel, "ctc", 4, 0.0, rng, B, T, L)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    tgt = rng.integers(0, nlabel - 1, size=(B, L)).astype(np.int32)
    xd = torch.tensor(x.reshape(B, nfeat, T)).cuda()
    td = torch.tensor(tgt).cuda()
    em_ref = ref.forward(x, params)
    em = tr.forward(xd, train=False).cpu().numpy()
    tr.forward_backward(xd, td)
    o = O.CTC(em_ref, tgt, scale_mode=4)
    o.forward()
    grads = ref.backward(o.backward().astype(np.float32), len(params))
    g = tr.grads.cpu().numpy()
    table = tr.param_table()
    errs = [rel(tr.export_from(i, g), want) for i, want in enumerate(grads)]
    print(f"[{tag}] seed {seed} stages {stages}: emission err {rel(em, em_ref):.1e}; gradient errors " +
          " ".join(f"{table[i][0]}:{e:.0e}" for i, e in enumerate(errs)))
    return errs


for stages in ([(10, 1, 2400)], [(18, 3, 4320)]):
    for seed in (21, 22, 23, 24, 25, 26):
        run("product (block-Toeplitz)", seed, stages)
        os.environ["W2L_TDS_TZ_OFF"] = "1"
        os.environ["W2L_TDS_TZF_OFF"] = "1"
        try:
            with _lib.use_probe():
                run("previous generation     ", seed, stages)
        finally:
            os.environ.pop("W2L_TDS_TZ_OFF"); os.environ.pop("W2L_TDS_TZF_OFF")
