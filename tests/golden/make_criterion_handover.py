"""Hand-over vector for the sequence criteria -- the part of the path whose reference implementation (Flashlight's
ForceAlignmentCriterion / FullConnectionCriterion / ViterbiPath / ConnectionistTemporalClassificationCriterion) is NOT in
the reference tree and holds no golden vector there ("parity unpinned").  A maintainer with Flashlight closes
the pin in a few lines:

    auto asg = fl::pkg::speech::ASGLoss(N, fl::lib::seq::CriterionScaleMode::NONE, 0.0);
    asg.setParams(fl::param(af::array(N, N, transitions.data())), 0);
    auto loss = asg.forward({fl::input(af::array(N, T, B, emissions.data())), fl::noGrad(af::array(L, B, target.data()))});
    auto path = asg.viterbiPath(af::array(N, T, B, emissions.data()));
    auto ctc  = fl::pkg::speech::CTCLoss(CriterionScaleMode::NONE).forward({emissions, ctc_target});   // blank = N - 1

Arrays are stored in ArrayFire MEMORY ORDER with their af dims: emissions (N, T, B) == memory [B][T][N]; target (L, B) ==
[B][L], -1 padded; transitions (N, N) memory [i][j] = score of moving FROM label j TO label i (the layout of this build's C
ABI and oracle).  The numbers come from oracle/criterion_oracle.c (fp64), NOT from Flashlight; the oracle
itself is pinned by brute-force path enumeration, finite differences and torch's ctc_loss (tests/test_oracle_criterion.py).
   python tests/golden/make_criterion_handover.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    B, T, N, L = 2, 9, 6, 4
    rng = np.random.default_rng(20260924)
    em = np.round(rng.normal(size=(B, T, N)), 3).astype(np.float32)
    A = np.round(rng.normal(size=(N, N)) * 0.5, 3).astype(np.float32)
    tgt = np.array([[2, 0, 3, 1], [4, 1, -1, -1]], np.int32)
    out = {"B": B, "T": T, "N": N, "L": L, "emissions_af_dims": [N, T, B], "emissions": [float(v) for v in em.reshape(-1)],
           "transitions_af_dims": [N, N], "transitions": [float(v) for v in A.reshape(-1)],
           "target_af_dims": [L, B], "target": [int(v) for v in tgt.reshape(-1)],
           "source": "oracle/criterion_oracle.c (fp64 restatement of SURVEY App. B); scale mode NONE"}
    fac = O.FAC(em, A, tgt)
    out["fac"] = [float(v) for v in fac.forward()]
    out["fac_viterbi_alignment"] = [int(v) for v in fac.viterbi().reshape(-1)]
    fcc = O.FCC(em, A, O.batch_target_size(tgt, T))
    out["fcc"] = [float(v) for v in fcc.forward()]
    loss, dx, dA = O.asg(em, A, tgt)
    out["asg_loss"] = [float(v) for v in loss]                      # fcc - fac
    out["asg_grad_emissions"] = [float(v) for v in np.asarray(dx).reshape(-1)]
    out["asg_grad_transitions"] = [float(v) for v in np.asarray(dA).reshape(-1)]
    out["viterbi_path"] = [int(v) for v in O.viterbi(em, A).reshape(-1)]
    ctc_t = np.where(tgt >= 0, np.minimum(tgt, N - 2), -1).astype(np.int32)   # labels 0..N-2, blank = N-1
    ctc = O.CTC(em, ctc_t)
    out["ctc_target"] = [int(v) for v in ctc_t.reshape(-1)]
    out["ctc_loss"] = [float(v) for v in ctc.forward()]
    out["ctc_grad_emissions"] = [float(v) for v in ctc.backward().reshape(-1)]
    out["ctc_viterbi_path"] = [int(v) for v in O.ctc_viterbi(em).reshape(-1)]
    out["tol"] = 1e-4
    json.dump(out, open(os.path.join(OUT, "criterion_handover.json"), "w"), indent=0)
    print("wrote criterion_handover.json: asg", out["asg_loss"], "ctc", out["ctc_loss"])


if __name__ == "__main__":
    main()
