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
