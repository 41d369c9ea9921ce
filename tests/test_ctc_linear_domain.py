"""CPU: the scaled linear-domain arithmetic of the HIP CTC scans (oracle/ctc_linear_domain.py: per-position fp64 mantissa + integer
exponent, p_t(s) through the integer / fraction split of exp, occupancies from the two scans) against the log-domain fp64 oracle
(oracle/criterion_oracle.c, held to torch's ctc_loss in tests/test_oracle_criterion.py): loss to 1e-6 relative, gradient to 2e-6
of its largest entry -- the fp32 row normaliser and the fp32 occupancy are the only roundings of note."""
import numpy as np
import pytest

from oracle import ctc_linear_domain as CL


def _case(rng, T, N, L, scale=2.0, repeats=False):
    x = (rng.normal(size=(T, N)) * scale).astype(np.float32)
    y = rng.integers(0, N - 1, size=L).astype(np.int32)
    if repeats and L >= 4:
        y[1] = y[0]; y[3] = y[2]
    return x, y


def _check(oracle, x, y, size=None, tol_loss=1e-6, tol_grad=2e-6, grad=1.0):
    T, N = x.shape
    L = max(1, len(y))
    tgt = np.full((1, L), -1, np.int32)
    tgt[0, :len(y)] = y
    ts = np.array([len(y) if size is None else size], np.int32)
    ctc = oracle.CTC(x[None], tgt, target_size=ts)
    want = ctc.forward()[0]
    dwant = ctc.backward(np.array([grad]))[0]
    got, dgot, st = CL.ctc_linear(x, y, ts[0], grad=grad)
    assert np.isfinite(want) and np.isfinite(got)
    assert abs(got - want) <= tol_loss * max(1.0, abs(want)), (got, want)
    assert np.abs(dgot - dwant).max() <= tol_grad * max(1e-30, np.abs(dwant).max()), np.abs(dgot - dwant).max()
    return st


@pytest.mark.parametrize("T,N,L,repeats", [(8, 5, 3, False), (30, 12, 6, True), (188, 60, 40, True), (64, 9, 0, False), (5, 4, 1, False)])
def test_linear_domain_scans_equal_the_log_domain_oracle(oracle, T, N, L, repeats):
    rng = np.random.default_rng(1000 + T + L)
    x, y = _case(rng, T, N, L, repeats=repeats)
    _check(oracle, x, y, grad=0.7)


def test_long_utterance_keeps_its_range_and_precision(oracle):
    """T = 1500 frames at near-uniform emissions over 400 classes: alpha decays to ~2^-13000, far below fp64 -- the per-position
    exponents carry it; loss and gradient stay at the short-utterance accuracy"""
    rng = np.random.default_rng(7)
    x, y = _case(rng, 1500, 400, 80, scale=1.0, repeats=True)
    st = _check(oracle, x, y)
    assert st["aE"][-1].max() < -5000            # (an fp64 value alone would have underflowed thousands of frames earlier)


def test_confident_wrong_frames_do_not_underflow(oracle):
    """logit gaps of 150-300 nats on the target's labels at some frames: exp in fp32 is 0 there (the round-3 kernel made such
    cells impossible); the integer / fraction split keeps p_t(s) as an fp64 value and the likelihood finite"""
    rng = np.random.default_rng(11)
    T, N, L = 40, 8, 18            # tight: 2 L + 1 = 37 positions in 40 frames, almost every frame must emit a label
    x, y = _case(rng, T, N, L)
    for t in (3, 11, 12, 29):
        x[t] = -300.0
        x[t, (y[min(t // 2, L - 1)] + 1) % (N - 1)] = 0.0      # all the mass on a wrong label
    lp = x - np.log(np.exp(x.astype(np.float64) - x.max(1, keepdims=True)).sum(1, keepdims=True)).astype(np.float32) - x.max(1, keepdims=True)
    assert (np.exp(lp.astype(np.float32)) == 0).any()          # fp32 exp underflows on these rows
    assert (CL.exp_wide(lp.min()) > 0)                          # the split does not
    _check(oracle, x, y, tol_loss=1e-6, tol_grad=5e-6)


def test_exact_fit_and_infeasible_targets(oracle):
    rng = np.random.default_rng(3)
    T, N = 6, 7
    x = rng.normal(size=(T, N)).astype(np.float32)
    y = np.array([0, 1, 2, 3, 4, 5], np.int32)                 # T == L, no repeats: one alignment
    _check(oracle, x, y)
    # a repeat needs a blank between: 3 labels with one repeat in 3 frames is infeasible when the size is forced
    loss, dx, _ = CL.ctc_linear(x[:3], np.array([1, 1, 2], np.int32), 3)
    assert loss == np.inf
    tgt = np.array([[1, 1, 2]], np.int32)
    assert oracle.CTC(x[None, :3], tgt, target_size=np.array([3], np.int32)).forward()[0] == np.inf      # as the log-domain oracle
    ghost, _, _ = CL.ctc_linear(x[:3], np.array([1, 1, 2], np.int32), 3, ghost=True)
    assert np.isfinite(ghost) and ghost > 1e8                  # what the scans returned before the forced exponent moved below NOEXP
    sm = np.exp(x[:3].astype(np.float64)); sm /= sm.sum(1, keepdims=True)
    assert np.abs(dx - sm).max() < 1e-6                        # the kernels' convention: no occupancy, gradient = softmax


def test_exp_wide_over_the_whole_range():
    lp = -np.concatenate([np.linspace(0, 120, 4001), np.linspace(120, 740, 2001)]).astype(np.float32)
    got = CL.exp_wide(lp)
    want = np.exp(lp.astype(np.float64))
    assert np.all(got > 0)                                      # down to -740: where fp32 exp gives 0 below -87 (-103)
    # the fp32 product lp * log2(e) carries a relative 6e-8: an ABSOLUTE error in the exponent that grows with |lp| -- p is good to
    # ~7e-8 (1 + |lp|): 4e-7 for a likely label, 7e-6 at lp = -100, 3e-5 at -300 (fp64 denormals below -700 excluded here)
    norm = lp > -700
    assert (np.abs(got / want - 1)[norm] / (1.0 + np.abs(lp[norm]))).max() < 1.5e-7
    assert CL.exp_wide(np.float32(-1e9)) == 0.0 and CL.exp_wide(np.float32(-740.0)) > 0   # below fp64's range p is 0, as exp() in fp64 is


@pytest.mark.parametrize("T,N,L", [(30, 12, 6), (188, 60, 40), (40, 8, 18)])
def test_ghost_mass_never_reached_a_feasible_result(oracle, T, N, L):
    """the fix above changes nothing for a feasible target: loss and gradient of the two formulations are bit-identical"""
    rng = np.random.default_rng(55 + T)
    x, y = _case(rng, T, N, L, repeats=True)
    ts = oracle.batch_ctc_target_size(y[None], T)[0]
    a = CL.ctc_linear(x, y, ts)
    b = CL.ctc_linear(x, y, ts, ghost=True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert (b[2]["aM"] > 0).sum() > (a[2]["aM"] > 0).sum()   # (the ghost positions existed)
