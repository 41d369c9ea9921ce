"""oracle/asg_linear_domain.py (the scaled linear-domain statement of the FCC / FAC recursions, the plan for the N = 30 criterion
kernels) against the log-domain oracle: losses, and the per-frame quantities the backward kernels read."""
import numpy as np
import pytest

from oracle import asg_linear_domain as LD
from oracle import pyoracle as O


def _case(T, N, L, S, scale, seed):
    rng = np.random.default_rng(seed)
    x = (rng.normal(size=(1, T, N)) * scale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.5 + np.eye(N)).astype(np.float32)
    tgt = np.full((1, L), -1, np.int32)
    tgt[0, :S] = rng.integers(0, N, S)
    return x, A, tgt


@pytest.mark.parametrize("T,N,scale", [(50, 6, 1.0), (400, 30, 1.0), (2000, 30, 1.0), (600, 30, 12.0), (300, 64, 3.0)])
def test_fcc_forward_in_the_linear_domain(T, N, scale):
    """loss within 2e-6 (fp32 chain) / 1e-10 (fp64 chain) of the oracle, relative; e_t stays normalised (max = 1), so nothing
    under- or overflows however large the emissions are (scale 12: log-partition ~ 3e4)"""
    x, A, tgt = _case(T, N, 8, 4, scale, T + N)
    want = O.FCC(x, A, np.array([4], np.int32)).forward()[0]
    got32, ahat, logs = LD.fcc_forward_linear(x[0], A, np.float32)
    got64, _, _ = LD.fcc_forward_linear(x[0], A, np.float64)
    assert abs(got64 - want) < 1e-10 * abs(want)
    assert abs(got32 - want) < 2e-6 * abs(want)
    assert np.isfinite(ahat).all() and np.isfinite(logs).all() and abs(ahat.max(axis=1)).max() < 1e-6


@pytest.mark.parametrize("T,N,L,S,scale", [(40, 6, 8, 5, 1.0), (500, 30, 100, 77, 1.0), (2000, 30, 300, 300, 1.0), (2000, 30, 300, 120, 8.0),
                                           (300, 30, 300, 299, 2.0), (64, 30, 40, 1, 1.0)])
def test_fac_forward_in_the_linear_domain(T, N, L, S, scale):
    """loss within 1e-9 relative of the oracle at the conv_glu shape (T = 2000, L = 300) and with large emissions, with an
    exponent per position and with one per lane of 5 adjacent positions (the planned kernel: 64 lanes x 5 >= 300); the stay
    weights w1 -- the only thing the backward scan reads -- are shares in [0, 1]"""
    x, A, tgt = _case(T, N, L, S, scale, T + S)
    f = O.FAC(x, A, tgt)
    want = f.forward()[0]
    for group in (1, 5):
        got, w1 = LD.fac_forward_linear(x[0], A, tgt[0], S, group)
        assert np.isfinite(got) and abs(got - want) < 1e-9 * max(1.0, abs(want)), (group, got, want)
        assert (w1 >= 0).all() and (w1 <= 1).all() and np.isfinite(w1).all()
    # input gradient of the oracle = occupancies: every frame's occupancies sum to 1 (a sanity anchor for the comparison above)
    dx, _ = f.backward()
    assert np.allclose(dx[0].sum(axis=1), 1.0, atol=1e-9)


def test_one_exponent_per_frame_is_not_enough_for_fac():
    """the negative result that shaped the plan: with ONE power-of-two scale per frame the lattice positions far behind the
    front underflow (1e-308 below the frame's largest entry) although the best path reaches them later -- the loss comes out
    hundreds of nats wrong at T = 2000 with large emissions, while the per-lane exponents stay exact"""
    x, A, tgt = _case(2000, 30, 300, 120, 8.0, 2120)
    want = O.FAC(x, A, tgt).forward()[0]
    bad, _ = LD.fac_forward_linear(x[0], A, tgt[0], 120, group=0)
    good, _ = LD.fac_forward_linear(x[0], A, tgt[0], 120, group=5)
    assert abs(good - want) < 1e-9 * abs(want)
    assert not np.isfinite(bad) or abs(bad - want) > 1.0


# ---- round 4: models of the arithmetic of the shipped kernels (csrc/criterion_asg_small.hip) ---------------------------------
@pytest.mark.parametrize("T,N,scale,tscale", [(50, 6, 1.0, 0.5), (400, 30, 1.0, 0.5), (2000, 30, 1.0, 0.5), (600, 30, 12.0, 0.5),
                                              (500, 31, 3.0, 4.0), (300, 30, 40.0, 8.0)])
def test_fcc_kernel_model_lagged_scale(T, N, scale, tscale):
    """the lagged power-of-two scale (exponent of the total mass two frames back, minus the correction already under way) keeps
    the fp32 chain in range for unit and for very large emissions / transition spreads; loss 2e-6, gradients 1e-4 of the
    largest entry (the parity bar), against the log-domain fp64 oracle"""
    rng = np.random.default_rng(T * 31 + N)
    x = (rng.normal(size=(1, T, N)) * scale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * tscale + np.eye(N) * 4).astype(np.float32)
    f = O.FCC(x, A, np.array([4], np.int32))
    want = f.forward()[0]
    got, u, q, ks = LD.fcc_kernel_model(x[0], A)
    assert np.isfinite(u).all() and abs(got - want) < 2e-6 * abs(want), (got, want)
    mass = u.astype(np.float64).sum(axis=1)
    assert mass.min() > 2.0 ** -100 and mass.max() < 2.0 ** 40, (mass.min(), mass.max())
    odx, odA = f.backward()
    dx, dA = LD.fcc_kernel_model_backward(u, q, A)
    assert np.abs(dx - odx[0]).max() < 1e-4 * np.abs(odx).max()
    assert np.abs(dA - odA).max() < 1e-4 * np.abs(odA).max()


@pytest.mark.parametrize("T,N,L,S,scale,P", [(40, 6, 8, 5, 1.0, 1), (500, 30, 100, 77, 1.0, 2), (2000, 30, 300, 300, 1.0, 5),
                                             (2000, 30, 300, 120, 8.0, 5), (300, 30, 300, 299, 2.0, 5), (64, 30, 40, 1, 1.0, 1),
                                             (700, 30, 64, 64, 30.0, 1), (900, 30, 320, 310, 25.0, 5), (301, 30, 300, 298, 22.0, 5), (301, 30, 300, 300, 22.0, 5),
                                             (304, 30, 300, 300, 20.0, 5), (200, 30, 64, 64, 22.0, 1)])
def test_fac_kernel_model_lane_exponents(T, N, L, S, scale, P):
    """fp64 mantissas, one exponent per lane of P positions, renormalised every 4 frames through the decaying maximum scan:
    loss 1e-6 relative (the per-frame factors are fp32 exp2 values), stay weights within 1e-5 of the oracle's wherever the
    backward pass can reach"""
    x, A, tgt = _case(T, N, L, S, scale, T + S + 7)
    f = O.FAC(x, A, tgt)
    want = f.forward()[0]
    got, w1 = LD.fac_kernel_model(x[0], A, tgt[0], S, P=P)
    assert np.isfinite(got) and abs(got - want) < 1e-6 * max(1.0, abs(want)), (got, want)
    assert (w1 >= 0).all() and (w1 <= 1.0 + 1e-6).all()
    # the backward recursion on the model's stay weights gives the oracle's input gradient (occupancies)
    dx, _ = f.backward()
    da = np.zeros(S)
    da[S - 1] = 1.0
    occ = np.zeros((T, N))
    y = tgt[0, :S]
    for t in range(T - 1, -1, -1):
        np.add.at(occ[t], y, da)
        if t >= 1:
            st = da * w1[t]
            adv = da - st
            da = st + np.concatenate((adv[1:], [0.0]))
    assert np.abs(occ - dx[0]).max() < 1e-4


def test_fac_kernel_model_flags_what_it_cannot_hold():
    """a tight alignment under emissions of scale 60 (frames that cost the forced path 400+ nats) is beyond one exponent per
    lane of 5 positions: the model returns a WRONG finite loss there -- and the quantity the kernel measures (largest per-frame
    spread + largest |log2 kappa|) is far above the bound at which it hands the utterance to the log-domain kernel; every case
    below the bound is exact"""
    x, A, tgt = _case(301, 30, 300, 298, 60.0, 301 + 298 + 7)
    want = O.FAC(x, A, tgt).forward()[0]
    got, _ = LD.fac_kernel_model(x[0], A, tgt[0], 298, P=5)
    assert LD.fac_kernel_gain_bits(x[0], A, tgt[0], 298) > 2 * LD.FAC_SAFE_BITS
    assert not np.isfinite(got) or abs(got - want) > 1e-3 * abs(want)
    # 40 random utterances at the bound (tight, loose, 1 / 2 / 5 positions per lane): all exact
    rng0 = np.random.default_rng(5)
    for it in range(40):
        S = int(rng0.integers(1, 320))
        T = max(2, S + int(rng0.choice([0, 1, 2, 3, 5, 8, 20, 60, 300])))
        P = 5 if S > 128 else int(rng0.choice([1, 2, 5])) if S <= 64 else int(rng0.choice([2, 5]))
        rng = np.random.default_rng(1000 + it)
        x = (rng.normal(size=(1, T, 30)) * float(rng0.choice([1, 5, 15, 19]))).astype(np.float32)
        A = (rng.normal(size=(30, 30)) * float(rng0.choice([0.1, 1.0, 3.0])) + np.eye(30) * 4).astype(np.float32)
        tgt = np.full((1, 320), -1, np.int32)
        tgt[0, :S] = rng.integers(0, 30, S)
        if LD.fac_kernel_gain_bits(x[0], A, tgt[0], S) > LD.FAC_SAFE_BITS:
            continue
        want = O.FAC(x, A, tgt).forward()[0]
        got, _ = LD.fac_kernel_model(x[0], A, tgt[0], S, P=P)
        assert np.isfinite(got) and abs(got - want) < 1e-6 * max(1.0, abs(want)) + 1e-7 * T, (it, T, S, P, got, want)   # (the factors are fp32 exp2 values: the error grows with T, not with the loss)
