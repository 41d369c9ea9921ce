"""TEST INFRASTRUCTURE (oracle side).  The ARITHMETIC of wav2letter_amd/csrc/criterion_ctc.hip's lattice scans restated in numpy, so that the
numerics of the scaled linear domain the kernels compute in are pinned on the CPU against the log-domain oracle (oracle/criterion_oracle.c's
CTC, itself held to torch's ctc_loss: tests/test_oracle_criterion.py) -- the CTC counterpart of oracle/asg_linear_domain.py.

What the kernels do, and this file repeats step for step:
  * ctc_rows_lse: lse_t in fp32; the label probabilities p_t(s) = exp(x_t[label(s)] - lse_t) as fp64 VALUES produced without an
    fp32 underflow -- `exp_wide`: z = lp * log2(e) (clamped at -1090), 2^(z - rint(z)) by the fp32 exp2, the integer part by ldexp;
  * ctc_scan: alpha_t(s) = (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2)) * p_t(s) with every lattice POSITION held as an
    fp64 mantissa in [0.5, 1) and its own integer exponent (kCtcNoExp = -2^28 for a position without mass): the three inputs are
    brought to their largest exponent with ldexp, summed, multiplied, renormalised with frexp; beta likewise from the last frame,
    INCLUDING p_t(s);  Z = alpha_{T-1}(S-1) + alpha_{T-1}(S-2) as zhat * 2^ez;
  * ctc_rows_grad: occupancy gamma_t(s) = alpha^ beta^ / (p zhat) * 2^(eA + eB - ez) (rounded to fp32), gradient
    g * (softmax(x_t) - sum over the positions of a label of gamma_t(s)).
Reference semantics: CTCLoss = ConnectionistTemporalClassificationCriterion, blank = N - 1 (recipes/slimIPL/src/Train.cpp:248-251,
:406-407); SURVEY App. B.4.  No reference code exists for this formulation: it is the builder's, which is why it is checked here."""
import numpy as np

NOEXP = -(1 << 28)
FORCED = 4096   # the exponent of a transition that does not exist sits this far BELOW an empty position's: see ctc_linear()


def exp_wide(lp):
    """fp32 log-probabilities (<= ~0) -> fp64 probabilities the way the row kernel forms them"""
    lp = np.asarray(lp, np.float32)
    z = np.maximum(lp * np.float32(1.44269504088896341), np.float32(-1090.0)).astype(np.float32)
    zi = np.rint(z).astype(np.float32)                    # v_rndne_f32: halves to even, as numpy
    frac = (z - zi).astype(np.float32)                    # in [-0.5, 0.5]
    m = np.exp2(frac.astype(np.float64)).astype(np.float32)   # v_exp_f32 (1 ulp)
    return np.ldexp(m.astype(np.float64), zi.astype(np.int64))


def _norm(h, E):
    m, fe = np.frexp(h)
    e = np.where(h > 0.0, E + fe, NOEXP)
    return m, e.astype(np.int64)


def _step(m, e, m1, e1, m2, e2, p):
    E = np.maximum(np.maximum(e, e1), e2)
    with np.errstate(under="ignore"):
        s = (np.ldexp(m, (e - E).clip(-5000, 0)) + np.ldexp(m1, (e1 - E).clip(-5000, 0))) + np.ldexp(m2, (e2 - E).clip(-5000, 0))
    return _norm(s * p, E)


def ctc_linear(x, target, size, grad=1.0, scale=1.0, ghost=False):
    """one utterance: x [T][N] fp32, target [L] (first `size` labels used), blank = N - 1.
    Returns loss (fp64), dx [T][N] (fp64, = scale * grad * d(-log Z)/dx), and the scan state for inspection.
    `ghost=True` reproduces the kernels before the round-4 fix: a disallowed skip transition carried the exponent NOEXP itself, so at a
    position fed by nothing else (E = NOEXP) its shift was 0 and the position inherited a mass of 2^-2^28 it cannot have."""
    x = np.asarray(x, np.float32)
    T, N = x.shape
    Lb = int(size)
    S = 2 * Lb + 1
    lab = np.full(S, N - 1, np.int64)
    lab[1::2] = np.asarray(target[:Lb], np.int64)
    x64 = x.astype(np.float64)
    mx = x64.max(axis=1)
    lse = (mx + np.log(np.exp(x64 - mx[:, None]).sum(axis=1))).astype(np.float32)   # (the kernel: fp32 block reductions)
    p = exp_wide((x[:, lab] - lse[:, None]).astype(np.float32))                       # [T][S] fp64
    s_idx = np.arange(S)
    skipA = (s_idx % 2 == 1) & (s_idx >= 2)
    skipA[skipA] = lab[s_idx[skipA]] != lab[s_idx[skipA] - 2]
    skipB = (s_idx % 2 == 1) & (s_idx + 2 < S)
    skipB[skipB] = lab[s_idx[skipB]] != lab[s_idx[skipB] + 2]
    zero, none = np.zeros(1), np.full(1, NOEXP, np.int64)

    def shift_up(m, e, k):      # value of position s - k at position s
        return np.concatenate([np.zeros(k), m[:S - k]])[:S], np.concatenate([np.full(k, NOEXP, np.int64), e[:S - k]])[:S]

    def shift_down(m, e, k):    # value of position s + k at position s
        return np.concatenate([m[k:], np.zeros(k)])[:S], np.concatenate([e[k:], np.full(k, NOEXP, np.int64)])[:S]

    aM = np.zeros((T, S)); aE = np.full((T, S), NOEXP, np.int64)
    bM = np.zeros((T, S)); bE = np.full((T, S), NOEXP, np.int64)
    aM[0], aE[0] = _norm(np.where(s_idx < 2, p[0], 0.0), 0)
    for t in range(1, T):
        m1, e1 = shift_up(aM[t - 1], aE[t - 1], 1)
        m2, e2 = shift_up(aM[t - 1], aE[t - 1], 2)
        e2 = np.where(skipA, e2, NOEXP - (0 if ghost else FORCED))
        aM[t], aE[t] = _step(aM[t - 1], aE[t - 1], m1, e1, m2, e2, p[t])
    bM[T - 1], bE[T - 1] = _norm(np.where(s_idx >= S - 2, p[T - 1], 0.0), 0)
    for t in range(T - 2, -1, -1):
        m1, e1 = shift_down(bM[t + 1], bE[t + 1], 1)
        m2, e2 = shift_down(bM[t + 1], bE[t + 1], 2)
        e2 = np.where(skipB, e2, NOEXP - (0 if ghost else FORCED))
        bM[t], bE[t] = _step(bM[t + 1], bE[t + 1], m1, e1, m2, e2, p[t])
    last = [s for s in (S - 1, S - 2) if s >= 0 and aM[T - 1, s] > 0.0]
    if not last:
        return np.inf, scale * grad * np.exp(x64 - lse[:, None].astype(np.float64)), dict(p=p, aM=aM, aE=aE, bM=bM, bE=bE)
    ez = max(int(aE[T - 1, s]) for s in last)
    zs = sum(float(np.ldexp(aM[T - 1, s], int(aE[T - 1, s]) - ez)) for s in last)
    ll = np.log(zs) + ez * 0.69314718055994530942
    g = scale * grad
    dx = g * np.exp((x - lse[:, None]).astype(np.float32)).astype(np.float64)         # g * softmax in fp32 arithmetic
    ok = (aM > 0.0) & (bM > 0.0)
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        gam = np.where(ok, np.ldexp(aM * bM / (p * zs), (aE + bE - ez).clip(-5000, 5000)), 0.0).astype(np.float32)
    for s in range(S):
        dx[:, lab[s]] -= g * gam[:, s].astype(np.float64)
    return -scale * ll, dx, dict(p=p, aM=aM, aE=aE, bM=bM, bE=bE, zs=zs, ez=ez)
