"""Round 4: the N <= 31 ASG kernels (csrc/criterion_asg_dpp.hpp: FCC and Viterbi on DPP row rotations in a scaled linear
domain; csrc/criterion_fac_lin.hpp: FAC with fp64 mantissas and one exponent per lane) and the widened CTC label
probabilities, against the CPU oracle through the C ABI.

Bar (BASELINE.json north_star): Viterbi paths bit-exact; loss / gradients within 1e-4 of the oracle relative to the largest
reference magnitude.  The cases walk the edges of the new machine mappings: state counts around the 16-lane row boundary and
up to 31, frame counts around the 16-frame chunk and its parity, target lengths around the positions-per-lane steps, emission
and transition magnitudes far beyond what fp32 exp() can hold (the scaled domains must not care)."""
import numpy as np
import pytest
import torch

from test_gpu_criterion import TOL, dev, gradrel, make_targets, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 29, 30, 31])
@pytest.mark.parametrize("T", [1, 2, 3, 15, 16, 17, 32, 33, 257])
def test_fcc_dpp_state_and_frame_edges(oracle, N, T):
    from wav2letter_amd import FullConnectionCriterion
    rng = np.random.default_rng(N * 1000 + T)
    B = 3
    x = (rng.normal(size=(B, T, N)) * 1.5).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = make_targets(rng, B, 5, N, T)
    w = rng.normal(size=B).astype(np.float32)
    crit = FullConnectionCriterion(N, 0).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    o = oracle.FCC(x, A, oracle.batch_target_size(tgt, T), 0)
    ol = o.forward()
    odx, odA = o.backward(w.astype(np.float64))
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    if T > 1:
        assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL
    else:
        assert np.abs(crit.transitions.grad.cpu().numpy()).max() == 0


@pytest.mark.parametrize("xscale,ascale,diag", [(1.0, 0.1, 4.0), (12.0, 0.5, 4.0), (40.0, 2.0, 0.0), (0.01, 8.0, 0.0), (3.0, 0.0, 0.0)])
def test_fcc_dpp_magnitudes(oracle, xscale, ascale, diag):
    """the lagged power-of-two scale: emissions of +-100 and more, transition rows spread over tens of nats, T = 2000 frames"""
    from wav2letter_amd import FullConnectionCriterion
    rng = np.random.default_rng(int(xscale * 10 + ascale * 100))
    B, T, N = 2, 2000, 30
    x = (rng.normal(size=(B, T, N)) * xscale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * ascale + np.eye(N) * diag).astype(np.float32)
    tgt = make_targets(rng, B, 5, N, T)
    crit = FullConnectionCriterion(N, 0).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    o = oracle.FCC(x, A, oracle.batch_target_size(tgt, T), 0)
    ol = o.forward()
    odx, odA = o.backward(np.ones(B))
    assert np.isfinite(loss.detach().cpu().numpy()).all()
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL


@pytest.mark.parametrize("L", [1, 2, 5, 63, 64, 65, 128, 129, 192, 193, 256, 257, 300, 320])
def test_fac_lin_positions_per_lane_edges(oracle, L):
    from wav2letter_amd import ForceAlignmentCriterion
    rng = np.random.default_rng(L)
    B, N = 3, 30
    T = max(L + 7, 40)
    x = (rng.normal(size=(B, T, N)) * 1.5).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = make_targets(rng, B, L, N, T)
    tgt[0, :L] = rng.integers(0, N, size=L)        # the whole width
    tgt[1, :] = -1
    tgt[1, 0] = 3                                  # a single label
    w = rng.normal(size=B).astype(np.float32)
    crit = ForceAlignmentCriterion(N, 4).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    o = oracle.FAC(x, A, tgt, scale_mode=4)
    ol = o.forward()
    odx, odA = o.backward(w.astype(np.float64))
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL


@pytest.mark.parametrize("T,L,xscale,ascale", [(2000, 300, 1.0, 0.1), (2000, 120, 8.0, 0.5), (900, 310, 22.0, 1.0), (700, 64, 22.0, 2.0),
                                               (300, 299, 2.0, 0.5), (301, 300, 22.0, 0.5), (304, 300, 20.0, 0.5), (330, 300, 15.0, 3.0),
                                               (301, 300, 60.0, 0.5), (304, 300, 60.0, 0.5), (1500, 7, 40.0, 3.0), (600, 200, 3.0, 12.0),
                                               (2000, 300, 1.0, 25.0), (600, 200, 3.0, 50.0), (2000, 64, 1.0, 100.0)])
def test_fac_lin_magnitudes(oracle, T, L, xscale, ascale):
    """one exponent per lane, renormalised every 4 frames: tight alignments (T ~ L: the lattice front IS the path), emissions
    whose per-frame spread exceeds what one fp32 (or one shared) scale can hold (scale 22: ~130 bits per frame, inside the
    range the linear-domain kernel keeps exact); beyond kFacSafeBits (scale 40, 60; transition spreads of tens of nats) the
    kernel flags the utterance and the log-domain kernel recomputes it -- the result must be right either way.  Transition
    rows 100+ nats wide (ascale 25 ... 100; round 5): kappa = exp(A[y_i][y_{i-1}] - A[y_{i-1}][y_{i-1}]) leaves the fp32 range --
    taken with __expf it was 0 or inf and the loss of such an utterance -inf / NaN (tools/exp/asg_wide_transitions.py)"""
    from wav2letter_amd import ForceAlignmentCriterion
    rng = np.random.default_rng(T + L)
    B, N = 2, 30
    x = (rng.normal(size=(B, T, N)) * xscale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * ascale + np.eye(N) * 4).astype(np.float32)
    tgt = make_targets(rng, B, L, N, T, min_len=max(1, L - 5))
    crit = ForceAlignmentCriterion(N, 0).cuda()
    crit.transitions.data = dev(A)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    loss.sum().backward()
    o = oracle.FAC(x, A, tgt, scale_mode=0)
    ol = o.forward()
    odx, odA = o.backward(np.ones(B))
    assert np.isfinite(loss.detach().cpu().numpy()).all()
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), odx) < TOL
    assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < TOL


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 30, 31])
@pytest.mark.parametrize("T", [1, 2, 16, 17, 129, 300])
def test_viterbi_dpp_bit_exact_edges(oracle, N, T):
    from wav2letter_amd import ASGLoss
    rng = np.random.default_rng(N * 77 + T)
    B = 3
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    crit = ASGLoss(N).cuda()
    crit.transitions.data = dev(A)
    assert (crit.viterbiPath(dev(x)).cpu().numpy() == oracle.viterbi(x, A)).all()
    # heavy ties (first maximum must win exactly like the CPU scan), and scores large enough that fp32 sums round
    xq = (np.round(x * 2) / 2).astype(np.float32)
    Aq = (np.round(A * 2) / 2).astype(np.float32)
    crit.transitions.data = dev(Aq)
    assert (crit.viterbiPath(dev(xq)).cpu().numpy() == oracle.viterbi(xq, Aq)).all()
    xb = (x * 1000).astype(np.float32)
    crit.transitions.data = dev(A)
    assert (crit.viterbiPath(dev(xb)).cpu().numpy() == oracle.viterbi(xb, A)).all()


@pytest.mark.parametrize("B,T,N,L,scale", [(2, 40, 30, 12, 60.0), (2, 25, 9998, 20, 40.0), (2, 300, 29, 140, 40.0)])
def test_ctc_wide_logit_gaps(oracle, B, T, N, L, scale):
    """labels 100+ nats below the frame's normaliser (round-3 advice): the scans multiply by exp(lp) as an fp64 value built from
    an integer / fraction split, so a confident-wrong frame keeps a finite probability; one utterance has T' == number of
    lattice steps it needs (every frame is forced)"""
    from wav2letter_amd import CTCLoss
    rng = np.random.default_rng(N + T)
    x = (rng.normal(size=(B, T, N)) * scale).astype(np.float32)
    tgt = make_targets(rng, B, L, N, T, hi=N - 1, no_adjacent_repeat=True)
    l0 = min(L, T)
    y0 = rng.integers(0, N - 1, size=l0)
    for i in range(1, l0):
        while y0[i] == y0[i - 1]:
            y0[i] = rng.integers(0, N - 1)
    tgt[0, :] = -1
    tgt[0, :l0] = y0
    w = rng.normal(size=B).astype(np.float32)
    crit = CTCLoss(0)
    xt = dev(x).requires_grad_(True)
    loss = crit(xt, dev(tgt))
    (loss * dev(w)).sum().backward()
    o = oracle.CTC(x, tgt, scale_mode=0)
    ol = o.forward()
    assert np.isfinite(ol).all() and np.isfinite(loss.detach().cpu().numpy()).all()
    assert relerr(loss.detach().cpu().numpy(), ol) < TOL
    assert gradrel(xt.grad.cpu().numpy(), o.backward(w.astype(np.float64))) < TOL


def test_asg_generations_agree_and_timing():
    """the probe library can run the previous kernel generation (W2L_ASG_OLD=1): both agree at the bench shape; prints the
    four kernel timings side by side (informational)"""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
from wav2letter_amd import _lib
_lib.use_probe().__enter__()
from wav2letter_amd import ASGLoss, CriterionScaleMode
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
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
loss = crit(x, tgt); loss.sum().backward()
out = {"loss": loss.detach().cpu().numpy().tolist(), "dx": x.grad.cpu().numpy().ravel()[::997].tolist(),
       "dA": crit.transitions.grad.cpu().numpy().ravel().tolist(), "path": crit.viterbiPath(x.detach()).cpu().numpy().ravel()[::13].tolist(),
       "fwd_ms": timeit(lambda: crit(x, tgt)), "fwd_bwd_ms": timeit(lambda: crit(x, tgt).sum().backward()),
       "fcc_ms": timeit(lambda: crit.fcc(x, tgt)), "fac_ms": timeit(lambda: crit.fac(x, tgt)), "vit_ms": timeit(lambda: crit.viterbiPath(x.detach()))}
print("RESULT" + json.dumps(out))
'''
    res = {}
    for name, env in (("new", {}), ("old", {"W2L_ASG_OLD": "1"})):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert p.returncode == 0, p.stderr[-2000:]
        import json
        res[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    n, o = res["new"], res["old"]
    print("ASG generations (ms): " + ", ".join(f"{k} new {n[k]:.3f} old {o[k]:.3f}" for k in ("fwd_ms", "fwd_bwd_ms", "fcc_ms", "fac_ms", "vit_ms")))
    assert np.abs(np.array(n["loss"]) - np.array(o["loss"])).max() < 1e-4 * np.abs(np.array(o["loss"])).max()
    assert np.abs(np.array(n["dx"]) - np.array(o["dx"])).max() < 1e-4 * np.abs(np.array(o["dx"])).max()
    assert np.abs(np.array(n["dA"]) - np.array(o["dA"])).max() < 1e-4 * np.abs(np.array(o["dA"])).max()
    assert n["path"] == o["path"]


_VARIANT_CODE = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from wav2letter_amd import _lib
_lib.use_probe().__enter__()
from oracle import pyoracle as O
from wav2letter_amd import ForceAlignmentCriterion, FullConnectionCriterion
from test_gpu_criterion import dev, gradrel, make_targets, relerr
for (B, T, N, L, xs) in [(3, 33, 30, 12, 1.5), (2, 700, 30, 300, 1.0), (2, 301, 29, 300, 6.0), (2, 257, 17, 130, 2.0), (2, 1, 5, 1, 1.0)]:
    rng = np.random.default_rng(T + L)
    x = (rng.normal(size=(B, T, N)) * xs).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = make_targets(rng, B, L, N, T, min_len=max(1, min(L, T) - 3))
    w = rng.normal(size=B).astype(np.float32)
    for cls, orc in ((FullConnectionCriterion, None), (ForceAlignmentCriterion, None)):
        crit = cls(N, 4).cuda(); crit.transitions.data = dev(A)
        xt = dev(x).requires_grad_(True)
        loss = crit(xt, dev(tgt)); (loss * dev(w)).sum().backward()
        o = O.FCC(x, A, O.batch_target_size(tgt, T), 4) if cls is FullConnectionCriterion else O.FAC(x, A, tgt, scale_mode=4)
        ol = o.forward(); odx, odA = o.backward(w.astype(np.float64))
        assert relerr(loss.detach().cpu().numpy(), ol) < 1e-4, (cls.__name__, T, L)
        assert gradrel(xt.grad.cpu().numpy(), odx) < 1e-4, (cls.__name__, T, L)
        if T > 1: assert gradrel(crit.transitions.grad.cpu().numpy(), odA) < 1e-4, (cls.__name__, T, L)
print("VARIANT OK")
'''


@pytest.mark.parametrize("env", [{"W2L_FCC_1WAVE": "1"}, {"W2L_FAC_GEN": "wave", "W2L_FAC_BWD": "wave"}, {"W2L_FAC_GEN": "blin", "W2L_FAC_BWD": "blk51"},
                                 {"W2L_FAC_GEN": "blin2", "W2L_FAC_BWD": "blk42"}, {"W2L_ASG_OLD": "1"}, {"W2L_FCC_DTRANS_OLD": "1"}, {"W2L_FAC_BWD32": "1"}, {"W2L_ASG_NOMITM": "1"}])
def test_asg_kernel_variants_of_the_probe_library(env):
    """the kernel generations the product does not run stay selectable in the probe library (A/B work) and stay correct: the
    one-wave FCC scans, the one-wave FAC scans (with their hand-over to the log-domain kernel), the barrier-per-frame FAC scans (rows by a sixth wave / from the pre-pass; backward 5 x 1 and 4 x 2),
    the round-3 log-domain kernels, the transition gradient by lane broadcasts instead of the MFMA kernel, the FAC backward scan with 32-frame chunks,
    the round-4 / round-5 full-length scans (fcc_*_dpp2, fac_*_plin) that the meet-in-the-middle pair replaced in round 6"""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", _VARIANT_CODE], env=e, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "VARIANT OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def _asg_composed(Lb, B, T, N, L, mode, x, tgt, A, w):
    """ASG as the composition of the two criterion calls (the sequence up to round 5): target sizes, FCC, FAC, three axpy"""
    from wav2letter_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    ts = torch.empty(B, dtype=torch.int32, device="cuda")
    loss = torch.empty(B, device="cuda"); loss2 = torch.empty(B, device="cuda")
    dx = torch.empty_like(x); dx2 = torch.empty_like(x)
    dt = torch.empty(N, N, device="cuda"); dt2 = torch.empty(N, N, device="cuda")
    wf = torch.empty(Lb.w2l_fcc_workspace_size(B, T, N), dtype=torch.uint8, device="cuda")
    wa = torch.empty(Lb.w2l_fac_workspace_size(B, T, N, L), dtype=torch.uint8, device="cuda")
    _lib.check(Lb.w2l_batch_target_size(B, L, T, tgt.data_ptr(), ts.data_ptr(), s))
    _lib.check(Lb.w2l_fcc_forward(B, T, N, mode, x.data_ptr(), ts.data_ptr(), A.data_ptr(), loss.data_ptr(), wf.data_ptr(), s))
    _lib.check(Lb.w2l_fac_forward(B, T, N, L, mode, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), A.data_ptr(), loss2.data_ptr(), wa.data_ptr(), s))
    _lib.check(Lb.w2l_axpy(loss.data_ptr(), loss2.data_ptr(), B, -1.0, s))
    _lib.check(Lb.w2l_fcc_backward(B, T, N, A.data_ptr(), w.data_ptr(), dx.data_ptr(), dt.data_ptr(), wf.data_ptr(), s))
    _lib.check(Lb.w2l_fac_backward(B, T, N, L, tgt.data_ptr(), ts.data_ptr(), w.data_ptr(), dx2.data_ptr(), dt2.data_ptr(), wa.data_ptr(), s))
    _lib.check(Lb.w2l_axpy(dx.data_ptr(), dx2.data_ptr(), B * T * N, -1.0, s))
    _lib.check(Lb.w2l_axpy(dt.data_ptr(), dt2.data_ptr(), N * N, -1.0, s))
    torch.cuda.synchronize()
    return loss, dx, dt


@pytest.mark.parametrize("B,T,N,L,mode,ascale", [(3, 57, 30, 20, 0, 0.3),      # the fused sequence (N <= 31, L <= 320)
                                                 (5, 257, 29, 300, 4, 0.3),    # five waves of positions, TARGET_SZ_SQRT
                                                 (4, 64, 32, 40, 0, 0.3),      # FAC fused, FCC on its 32-state kernels
                                                 (2, 700, 30, 120, 2, 60.0),   # every utterance flagged: the log-domain recomputation inside the finish launch
                                                 (3, 40, 40, 25, 0, 0.3),      # N > 32: the composed calls behind the same entry points
                                                 (2, 33, 30, 330, 0, 0.3)])    # L > 320: likewise
def test_asg_in_one_call_equals_the_composed_calls_and_the_oracle(oracle, B, T, N, L, mode, ascale):
    """w2l_asg_forward / w2l_asg_backward (what fl::pkg::speech::ASGLoss and the trainer enqueue): bit-identical to the composition
    of w2l_fcc_* and w2l_fac_* (same operations, same operands, same order), 1e-4 against the fp64 oracle; with an empty target in
    the batch, targets longer than T (capped), a second backward on the same forward, and two workspaces interleaved"""
    from wav2letter_amd import _lib
    Lb = _lib.lib()
    rng = np.random.default_rng(B * 100 + T + N)
    x = dev((rng.normal(size=(B, T, N)) * 1.5).astype(np.float32))
    A = dev((rng.normal(size=(N, N)) * ascale + np.eye(N) * 2.0).astype(np.float32))
    tg = make_targets(rng, B, L, N, T)
    tg[0, :] = -1                                 # an utterance without a transcription
    if L > T: tg[1, :] = rng.integers(0, N, L)    # longer than the utterance: the target size is capped at T
    tgt = dev(tg)
    wv = rng.normal(size=B).astype(np.float32)
    w = dev(wv)
    s = torch.cuda.current_stream().cuda_stream
    want = _asg_composed(Lb, B, T, N, L, mode, x, tgt, A, w)

    def run(ws, second_backward=False):
        loss = torch.empty(B, device="cuda"); dx = torch.empty_like(x); dt = torch.empty(N, N, device="cuda")
        _lib.check(Lb.w2l_asg_forward(B, T, N, L, mode, x.data_ptr(), tgt.data_ptr(), A.data_ptr(), loss.data_ptr(), ws.data_ptr(), s))
        return loss, dx, dt

    def bwd(ws, dx, dt):
        _lib.check(Lb.w2l_asg_backward(B, T, N, L, tgt.data_ptr(), A.data_ptr(), w.data_ptr(), dx.data_ptr(), dt.data_ptr(), ws.data_ptr(), s))
    nbytes = Lb.w2l_asg_workspace_size(B, T, N, L)
    assert nbytes > 0
    ws1 = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); ws2 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    loss, dx, dt = run(ws1)
    bwd(ws1, dx, dt)
    torch.cuda.synchronize()
    for got, ref in zip((loss, dx, dt), want):
        assert torch.equal(got, ref)
    bwd(ws1, dx, dt)                              # a second backward on the same forward
    torch.cuda.synchronize()
    assert torch.equal(dx, want[1]) and torch.equal(dt, want[2])
    l1, dx1, dt1 = run(ws1); l2, dx2, dt2 = run(ws2)   # two forwards pending, backwards in the other order
    bwd(ws2, dx2, dt2); bwd(ws1, dx1, dt1)
    torch.cuda.synchronize()
    for got in ((l1, dx1, dt1), (l2, dx2, dt2)):
        for g_, ref in zip(got, want):
            assert torch.equal(g_, ref)
    ol, odx, odt = oracle.asg(x.cpu().numpy(), A.cpu().numpy(), tg, mode, wv.astype(np.float64))
    assert relerr(loss.cpu().numpy(), ol) < TOL
    assert gradrel(dx.cpu().numpy(), odx) < TOL
    assert gradrel(dt.cpu().numpy(), odt) < TOL
