"""Oracle parity AT THE SHAPES THE BENCH REPORTS NUMBERS ON (round-1 verdict, "what's weak" 1):

  * ASG / FCC / FAC / Viterbi at the north-star label width N = 9998 with random transitions,
  * the fp32 exp-domain FCC recursion over T = 1500 dependent steps against the fp64 oracle,
  * ASG forward + backward at the full BASELINE config-4 criterion shape (64 x 2000 x 30, L <= 300),
  * the reference's TDSBlock golden vector through the HIP TDS operators,
  * fl::SpecAugment: bit-exact against the oracle restatement + mask properties.

Bar (BASELINE.json north_star): integer outputs bit-exact; loss / gradients within 1e-4 of the
fp64 oracle relative to the largest reference magnitude.  The criteria's arithmetic is un-vendored
Flashlight, so the oracle itself is "parity unpinned" (oracle/criterion_oracle.c header).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import refnet

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def relerr(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    return np.abs(got - want).max() / max(1.0, np.abs(want).max())


def gradrel(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    return np.abs(got - want).max() / max(1e-30, np.abs(want).max())


def asg_targets(rng, B, L, nlet, lo, hi):
    """letter targets without identical neighbours (the replabel convention of the ASG recipes)"""
    tgt = np.full((B, L), -1, np.int32)
    for b in range(B):
        l = int(rng.integers(lo, hi + 1))
        y = rng.integers(0, nlet, size=l)
        for i in range(1, l):
            if y[i] == y[i - 1]:
                y[i] = (y[i] + 1) % nlet
        tgt[b, :l] = y
    return tgt


# ----------------------------------------------------------------------------------------------
# N = 9998 (criterion_fcc_big.hip, fac on word pieces, vit_big_*)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 4])
def test_asg_n9998_random_transitions_matches_oracle(oracle, mode):
    """FCC, FAC and ASG forward/backward at the stress width with NON-trivial transitions
    (B = 2, T = 8: 1.6e9 log-sum-exp terms for the oracle)"""
    from wav2letter_amd import ASGLoss, ForceAlignmentCriterion, FullConnectionCriterion
    rng = np.random.default_rng(9998 + mode)
    B, T, N, L = 2, 8, 9998, 6
    x = (rng.normal(size=(B, T, N)) * 1.5).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.5 + 4.0 * np.eye(N)).astype(np.float32)
    tgt = np.full((B, L), -1, np.int32)
    tgt[0, :5] = rng.integers(0, N, 5)
    tgt[1, :3] = rng.integers(0, N, 3)
    w = rng.uniform(0.5, 1.5, size=B).astype(np.float32)
    ts = oracle.batch_target_size(tgt, T)
    Ad = dev(A)

    fcc = FullConnectionCriterion(N, mode).cuda()
    fcc.transitions.data = Ad.clone()
    xt = dev(x).requires_grad_(True)
    loss = fcc(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    o = oracle.FCC(x, A, ts, mode)
    ol = o.forward()
    odx, odA = o.backward(w.astype(np.float64))
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(fcc.transitions.grad.cpu().numpy(), odA) < TOL
    fcc_l, fcc_dx, fcc_dA = ol, odx, odA
    del fcc, o

    fac = ForceAlignmentCriterion(N, mode).cuda()
    fac.transitions.data = Ad.clone()
    xt = dev(x).requires_grad_(True)
    loss = fac(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    o = oracle.FAC(x, A, tgt, scale_mode=mode)
    ol = o.forward()
    odx, odA = o.backward(w.astype(np.float64))
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(fac.transitions.grad.cpu().numpy(), odA) < TOL
    assert (fac.viterbiPath(dev(x), dev(tgt)).cpu().numpy() == o.viterbi()).all()
    del fac

    asg = ASGLoss(N, mode, 0.0).cuda()
    asg.transitions.data = Ad.clone()
    xt = dev(x).requires_grad_(True)
    loss = asg(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    assert relerr(loss.detach().cpu().numpy(), fcc_l - ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), fcc_dx - odx) < TOL
    assert gradrel(asg.transitions.grad.cpu().numpy(), fcc_dA - odA) < TOL
    # ViterbiPath at the same width: bit-exact
    assert (asg.viterbiPath(dev(x)).cpu().numpy() == oracle.viterbi(x, A)).all()


def test_fcc_long_recursion_t1500_matches_fp64_oracle(oracle):
    """fp32 exp-domain rescaling over 1500 DEPENDENT steps (the stress leg's T) against the fp64
    log-domain oracle: B = 1, T = 1500, N = 1000 through criterion_fcc_big.hip"""
    from wav2letter_amd import FullConnectionCriterion
    rng = np.random.default_rng(1500)
    B, T, N = 1, 1500, 1000
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.1 + 4.0 * np.eye(N)).astype(np.float32)
    tgt = np.zeros((B, 8), np.int32)
    crit = FullConnectionCriterion(N, 4).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    o = oracle.FCC(x, A, oracle.batch_target_size(tgt, T), 4)
    ol = o.forward()
    odx, odA = o.backward()
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL


def test_fcc_n9998_beta_pass_and_transition_gradient_t400(oracle):
    """The beta pass and the transition gradient AT the stress width (round-4 verdict, weak 3: they were held to the
    oracle at N = 9998 over T = 8 and at N = 1000 over T = 1500 only): one utterance, T = 400 dependent frames, N = 9998,
    random transitions -- 399 launches of the transposed-stream kernel + fcc_big_bwd_step, then the dA product over (t, b)
    -- against the fp64 log-domain oracle (4e10 log-sum-exp terms forward, twice that backward: host threads as the
    bench's check uses them)."""
    from wav2letter_amd import FullConnectionCriterion
    rng = np.random.default_rng(400)
    B, T, N = 1, 400, 9998
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.1 + 4.0 * np.eye(N)).astype(np.float32)
    tgt = np.zeros((B, 8), np.int32)
    crit = FullConnectionCriterion(N, 4).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    got_l = loss.detach().cpu().numpy()
    got_dx = xt.grad.cpu().numpy()
    got_dA = crit.transitions.grad.cpu().numpy()
    del crit, xt, loss
    torch.cuda.empty_cache()
    oracle.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))   # (as bench.py's host_threads(): more than 64 OpenMP threads run slower)
    o = oracle.FCC(x, A, oracle.batch_target_size(tgt, T), 4)
    ol = o.forward()
    odx, odA = o.backward()
    assert relerr(got_l, ol) < TOL
    assert gradrel(got_dx, odx) < TOL
    assert gradrel(got_dA, odA) < TOL
    # every frame's emission gradient is a distribution over the labels scaled by the criterion's scale: none is skipped
    s = got_dx.reshape(T, N).sum(axis=1)
    assert np.abs(s - s[0]).max() < 1e-3 * abs(s[0])


def test_asg_long_recursion_small_labels_t1500(oracle):
    """the same 1500-step recursion on the N <= 64 single-launch scans (fp64 offsets), ASG with long targets"""
    from wav2letter_amd import ASGLoss
    rng = np.random.default_rng(1501)
    B, T, N, L = 3, 1500, 30, 300
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = (np.eye(N) * 4 + rng.normal(size=(N, N)) * 0.1).astype(np.float32)
    tgt = asg_targets(rng, B, L, 28, 60, L)
    crit = ASGLoss(N, 4, 4.0).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    ol, odx, odA = oracle.asg(x, A, tgt, 4)
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL


# ----------------------------------------------------------------------------------------------
# BASELINE config 4 criterion shape, full size
# ----------------------------------------------------------------------------------------------
def test_asg_full_config4_shape_matches_oracle(oracle):
    """ASG forward + backward at B = 64, T = 2000, N = 30, L <= 300 (--transdiag=4, target/sqrt scaling) against the
    oracle, element by element: loss [64], emission gradient [64][2000][30], transition gradient [30][30]; forced
    alignment and Viterbi paths bit-exact at the same shape"""
    from wav2letter_amd import ASGLoss, CriterionScaleMode
    rng = np.random.default_rng(64)
    B, T, N, L = 64, 2000, 30, 300
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = (np.eye(N) * 4 + rng.normal(size=(N, N)) * 0.1).astype(np.float32)
    tgt = asg_targets(rng, B, L, 28, 60, L)
    crit = ASGLoss(N, CriterionScaleMode.TARGET_SZ_SQRT, 4.0).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    ol, odx, odA = oracle.asg(x, A, tgt, 4)
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL
    fac = oracle.FAC(x, A, tgt, scale_mode=4)
    fac.forward()
    assert (crit.viterbiPathWithTarget(dev(x), dev(tgt)).cpu().numpy() == fac.viterbi()).all()


# ----------------------------------------------------------------------------------------------
# the reference's own TDSBlock golden vector through the HIP operators
# ----------------------------------------------------------------------------------------------
def test_golden_tdsblock_through_hip_path():
    """recipes/streaming_convnets/inference/inference/module/test/TDSBlockTest.cpp:27-188 (committed as
    tests/golden/tdsblock_golden.json by tests/golden/make_golden.py): conv + ReLU + residual + per-frame LayerNorm
    + two Linear + residual + LayerNorm on the device, streaming LayerNorm form (no epsilon), against the
    reference's expected output at the reference's own tolerance"""
    from wav2letter_amd import ops
    g = json.load(open(os.path.join(GOLD, "tdsblock_golden.json")))
    T, H, Cc, kw = g["T"], g["groups"], g["channels"] // g["groups"], g["kernelSize"]
    # reference input is frame-major [T][H][C] already == device layout [B=1][T][H][C]
    xd = dev(np.array(g["in"], np.float32).reshape(1, T, H, Cc))
    wc_ref = np.array(g["conv_weights"], np.float32).reshape(Cc, kw, Cc)       # [co][k][ci]
    wc = dev(np.ascontiguousarray(wc_ref.transpose(1, 2, 0)))                  # [k][ci][co]
    bc = dev(np.array(g["conv_bias"], np.float32))
    gb1 = dev(np.array([g["ln1_weights"][0], g["ln1_bias"][0]], np.float32))
    gb2 = dev(np.array([g["ln2_weights"][0], g["ln2_bias"][0]], np.float32))
    l = H * Cc
    w1 = dev(np.array(g["lin1_weights"], np.float32).reshape(l, l))            # memory [in][out]
    b1 = dev(np.array(g["lin1_bias"], np.float32))
    w2 = dev(np.array(g["lin2_weights"], np.float32).reshape(l, l))
    b2 = dev(np.array(g["lin2_bias"], np.float32))
    a = ops.conv_forward(xd, wc, bc, 1, g["leftPadding"], g["rightPadding"], relu=True)
    y, _, _ = ops.residual_layernorm_forward(a, xd, gb1, T, eps=0.0)          # groups = frames: LN over (H, C)
    u = ops.linear_forward(y.view(T, l), w1, b1, relu=True)
    v = ops.linear_forward(u, w2, b2)
    out, _, _ = ops.residual_layernorm_forward(v.view(1, T, H, Cc), y, gb2, T, eps=0.0)
    err = np.abs(out.cpu().numpy().reshape(-1) - np.array(g["expectedOutput"])).max()
    assert err < g["tol"], err
    assert err < 2e-3, err  # the restatement reproduces it to 6e-4 (fp16-packed weights in the reference)


# ----------------------------------------------------------------------------------------------
# fl::SpecAugment
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,F,args", [(32, 1500, 80, (27, 2, 100, 1.0, 2)),     # SAUG 80 27 2 100 1.0 2
                                        (3, 57, 40, (15, 1, 50, 0.2, 2)),
                                        (2, 9, 8, (8, 2, 100, 1.0, 1)),
                                        (2, 300, 80, (0, 2, 0, 1.0, 2))])         # nothing to mask
def test_specaugment_bit_exact_and_mask_properties(oracle, B, T, F, args):
    from wav2letter_amd import _lib
    fmf, nf, tmt, tmp_, nt = args
    L = _lib.lib()
    rng = np.random.default_rng(T + F)
    x = rng.normal(size=(B, T, F)).astype(np.float32)
    x[x == 0] = 1.0
    seen = set()
    for seed in (1, 2, 3, 0xDEADBEEF):
        xd = dev(x)
        _lib.check(L.w2l_specaugment_inplace(xd.data_ptr(), B, T, F, fmf, nf, tmt, tmp_, nt, seed, None), "specaug")
        torch.cuda.synchronize()
        got = xd.cpu().numpy()
        want, m = oracle.specaugment(x, fmf, nf, tmt, tmp_, nt, seed)
        assert (got == want).all()
        zero = got == 0
        # the same masks for every utterance of the batch (af::span over the batch dim)
        assert (zero == zero[0:1]).all()
        # rows / columns are masked whole; everything else is untouched
        fm = zero[0].all(axis=0)
        tm = zero[0].all(axis=1)
        assert (zero[0] == (fm[None, :] | tm[:, None])).all()
        assert (got[~zero] == x[~zero]).all()
        # bounds: each frequency mask covers 1 .. fMaskF channels, each time mask 1 .. min(tMaskT, T*p) frames
        tmax = min(tmt, int(T * tmp_), T)
        assert fm.sum() <= nf * fmf and tm.sum() <= nt * tmax
        if fmf > 0:
            assert fm.sum() >= 1
            for k in range(nf):
                assert 0 <= m[0, k] <= m[1, k] < F and m[1, k] - m[0, k] < fmf
        if tmax > 0:
            assert tm.sum() >= 1
            for k in range(nt):
                assert 0 <= m[2, k] <= m[3, k] < T and m[3, k] - m[2, k] < tmax
        seen.add(got.tobytes())
        # deterministic per seed
        xd2 = dev(x)
        L.w2l_specaugment_inplace(xd2.data_ptr(), B, T, F, fmf, nf, tmt, tmp_, nt, seed, None)
        assert torch.equal(xd, xd2)
    if fmf > 0:
        assert len(seen) > 1  # different seeds draw different masks


def test_specaugment_rejects_narrow_input():
    """the reference throws when the input has fewer frequency channels than the mask width"""
    from wav2letter_amd import _lib
    x = torch.zeros(1, 10, 8, device="cuda")
    assert _lib.lib().w2l_specaugment_inplace(x.data_ptr(), 1, 10, 8, 27, 2, 100, 1.0, 2, 1, None) == _lib.W2L_EINVAL


def test_golden_transformer_block_through_hip_path():
    """tests/golden/transformer_block_golden.json -- the hand-over vector for a reference-side check of the Transformer block
    (generator: tests/golden/make_transformer_golden.py) -- through the C++ host graph and the HIP attention path: parameters
    imported in the reference's order and ArrayFire layouts, output with and without the padding mask"""
    from wav2letter_amd.trainer import Trainer
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "transformer_block_golden.json")))
    C, T, B = g["modelDim"], g["T"], g["B"]
    tr = Trainer("V -1 1 NFEAT 0\nRO 2 0 3 1\n" + g["arch_line"] + "\n", C, C, "ctc", 4)
    table = tr.param_table()
    assert len(table) == len(g["params"]) == 15
    for i, p in enumerate(g["params"]):
        assert table[i][1] == len(p["data"]), (table[i], p["name"])
        tr.import_param(i, np.array(p["data"], np.float32))
    tr.plan(B, T, 2)
    tr.to_device()
    x = np.array(g["x"], np.float32).reshape(B, T, C)
    xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()          # features (T, NFEAT, 1, B) == [B][NFEAT][T]
    y = tr.forward(xd, train=False).cpu().numpy()
    assert relerr(y.reshape(-1), np.array(g["y"])) < g["tol"]
    tr.set_input_sizes(torch.tensor(g["input_sizes"], dtype=torch.float32).cuda())
    ym = tr.forward(xd, train=False).cpu().numpy()
    assert relerr(ym.reshape(-1), np.array(g["y_masked"])) < g["tol"]
    assert relerr(ym.reshape(-1), np.array(g["y"])) > 1e-3


def test_golden_criterion_handover_through_hip_path():
    """tests/golden/criterion_handover.json -- the hand-over vector for a reference-side check of ASG (FAC, FCC, Viterbi) and
    CTC -- through the HIP criteria: losses and gradients within 1e-4, Viterbi paths and the forced alignment bit-exact"""
    from wav2letter_amd import ASGLoss, CTCLoss
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "criterion_handover.json")))
    B, T, N, L = g["B"], g["T"], g["N"], g["L"]
    em = np.array(g["emissions"], np.float32).reshape(B, T, N)
    A = np.array(g["transitions"], np.float32).reshape(N, N)
    tgt = np.array(g["target"], np.int32).reshape(B, L)
    asg = ASGLoss(N).cuda()
    asg.transitions.data = dev(A)
    x = dev(em).requires_grad_(True)
    loss = asg(x, dev(tgt))
    loss.sum().backward()
    assert relerr(loss.detach().cpu().numpy(), g["asg_loss"]) < g["tol"]
    assert gradrel(x.grad.cpu().numpy().reshape(-1), g["asg_grad_emissions"]) < g["tol"]
    assert gradrel(asg.transitions.grad.cpu().numpy().reshape(-1), g["asg_grad_transitions"]) < g["tol"]
    assert (asg.viterbiPath(dev(em)).cpu().numpy().reshape(-1) == np.array(g["viterbi_path"])).all()
    assert (asg.viterbiPathWithTarget(dev(em), dev(tgt)).cpu().numpy().reshape(-1) == np.array(g["fac_viterbi_alignment"])).all()
    ct = np.array(g["ctc_target"], np.int32).reshape(B, L)
    xc = dev(em).requires_grad_(True)
    ctc = CTCLoss()
    lc = ctc(xc, dev(ct))
    lc.sum().backward()
    assert relerr(lc.detach().cpu().numpy(), g["ctc_loss"]) < g["tol"]
    assert gradrel(xc.grad.cpu().numpy().reshape(-1), g["ctc_grad_emissions"]) < g["tol"]
    assert (ctc.viterbiPath(dev(em)).cpu().numpy().reshape(-1) == np.array(g["ctc_viterbi_path"])).all()
