"""Interchange vector for ONE Transformer block (arch token `TR 16 24 2 5 0.0 0.0`), written so that a maintainer with
Flashlight can pin this build against the real fl::Transformer / TransformerCPC:

    auto tr = std::make_shared<w2l::cpc::TransformerCPC>(16, 8, 24, 2, 5, 0.0f, 0.0f, false, false);
    for (int i = 0; i < 17; ++i) tr->setParams(fl::param(af::array(<dims>, <data>)), i);   // norm entries: weight, then bias
    tr->eval();  auto y = tr->forward({fl::input(af::array(af::dim4(16, 7, 2), x.data())), padMask}).front();

Every array is stored in ArrayFire (column-major) MEMORY ORDER with its af dims, i.e. exactly the buffer af::array(dims, ptr)
takes: x / y (C, T, B); position table (2*csz-1, d); Linear weights (out, in); biases (out); a LayerNorm entry holds the
module's TWO scalar parameters (weight, bias), each of af dims (1) -- 17 fl parameters in 15 entries.
`y` is the output without a padding mask; `y_masked` with forwardSequentialModuleWithPadMask's mask for `input_sizes`
(cpc/SequentialBuilder.cpp:58-81).  The reference tree holds NO vector for this module: these numbers come from
oracle/transformer_oracle.py (torch float64), NOT from Flashlight -- the file pins the oracle and the HIP path against each
other and against regressions, and is the hand-over point for a reference-side check.   python tests/golden/make_transformer_golden.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import transformer_oracle as TO  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    C, mlp, H, csz, T, B = 16, 24, 2, 5, 7, 2
    rng = np.random.default_rng(20260923)
    shapes = TO.tr_param_shapes(C, mlp, H, csz)
    names = ["posemb", "w1.weight", "w1.bias", "w2.weight", "w2.bias", "wq.weight", "wq.bias", "wk.weight", "wk.bias",
             "wv.weight", "wv.bias", "wf.weight", "wf.bias", "norm1.weight+bias", "norm2.weight+bias"]
    params, meta = [], []
    for name, (kind, shp) in zip(names, shapes):
        if kind == "ln":
            a = np.array([1 + 0.1 * rng.normal(), 0.1 * rng.normal()], np.float32)
            af_dims = [1]
        else:
            a = np.round(rng.uniform(-0.5, 0.5, size=shp), 4).astype(np.float32)
            af_dims = list(shp[::-1])     # numpy (memory-order) shape reversed == ArrayFire dims
        params.append(a)
        meta.append({"name": name, "af_dims": af_dims, "data": [float(v) for v in a.reshape(-1)]})
    x = np.round(rng.normal(size=(B, T, C)), 4).astype(np.float32)          # memory [B][T][C] == af (C, T, B)
    pt = [torch.tensor(p, dtype=torch.float64) for p in params]
    xt = torch.tensor(x, dtype=torch.float64)
    y = TO.tr_block(xt, pt, H, csz).numpy()
    sizes = [7000.0, 3100.0]
    kl = TO.key_lengths(sizes, T, T)
    ym = TO.tr_block(xt, pt, H, csz, key_len=kl).numpy()
    out = {"arch_line": f"TR {C} {mlp} {H} {csz} 0.0 0.0", "modelDim": C, "mlpDim": mlp, "nHeads": H, "csz": csz, "T": T, "B": B,
           "source": "oracle/transformer_oracle.py (restating recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:41-182)",
           "params": meta, "x_af_dims": [C, T, B], "x": [float(v) for v in x.reshape(-1)],
           "y": [float(v) for v in y.reshape(-1)], "input_sizes": sizes, "key_lengths": [int(v) for v in kl],
           "y_masked": [float(v) for v in ym.reshape(-1)], "tol": 1e-4}
    json.dump(out, open(os.path.join(OUT, "transformer_block_golden.json"), "w"), indent=0)
    print("wrote transformer_block_golden.json:", sum(len(m["data"]) for m in meta), "parameter floats")


if __name__ == "__main__":
    main()
