"""Scaled LINEAR-domain restatements of the two ASG recursions -- test infrastructure (oracle/), not product code.

Why: at the conv_glu shape (N = 30 tokens, T = 2000 frames) the criterion kernels are chains of dependent operations per frame
(FCC ~560 cycles, FAC ~580), and a good part of each chain is the exp / log pair of the log-domain update.  CTC went
from 0.36 to 0.16 ms when its lattice moved to a scaled linear domain (criterion_ctc.hip, round 3).  This file states the same
move for FullConnectionCriterion and ForceAlignmentCriterion in numpy, with the arithmetic types a kernel would use, so that
the numerics (what the per-frame rescaling keeps, what underflows, how the quantities the backward passes consume are
recovered) are pinned against the log-domain oracle BEFORE a kernel is written: tests/test_asg_linear_domain.py.

Reference semantics: Flashlight's cpu/{FullConnection,ForceAlignment}Criterion (un-vendored; restated in oracle/criterion_oracle.c,
called at recipes/slimIPL/src/Train.cpp:408-410).

FCC (one utterance, emissions x [T][N], transitions A[to][from]):
    log domain     alpha_t[j] = x_t[j] + lse_i(alpha_{t-1}[i] + A[j][i])
    linear domain  E[j][i] = exp(A[j][i] - rowmax_j)                      once
                   px_t[j] = exp(x_t[j] + rowmax_j - m_t),  m_t = max_j(x_t[j] + rowmax_j)      OFF the chain (x is known)
                   v = (E e_{t-1}) * px_t ;  e_t = v / max(v) ;  C += m_t + log max(v)          the chain: mat-vec, multiply, max
    loss = C + log sum_j e_T[j].  ahat_t = log e_t and log s_t (s = E e_{t-1}) -- what fcc_bwd_small reads -- are logs of chain
    values, taken off the chain.
FAC (target y[0..S), positions i):
    log domain     alpha_t[i] = x_t[y_i] + lse(alpha_{t-1}[i] + A[y_i][y_i], alpha_{t-1}[i-1] + A[y_i][y_{i-1}])
    linear domain  a_t[i] = (a_{t-1}[i] eS[i] + a_{t-1}[i-1] eP[i]) * ex_t[i],  ex_t[i] = exp(x_t[y_i] - m_t), m_t = max_i x_t[y_i]
                   fp64 mantissas with a power-of-two exponent per LANE (the P adjacent positions a lane holds), renormalised every
                   frame (exact); the neighbouring lane's exponent travels with its value.  ONE exponent per frame is not enough
                   (see fac_forward_linear).  w1_t[i] = a_{t-1}[i] eS[i] / (sum) is a plain quotient.
"""
import numpy as np


def fcc_forward_linear(x, trans, dtype=np.float32):
    """x [T][N], trans [N][N] (to, from).  Returns (loss, ahat [T][N], logs [T][N]) with chain arithmetic in `dtype`."""
    x = np.asarray(x, np.float64)
    A = np.asarray(trans, np.float64)
    T, N = x.shape
    rowmax = A.max(axis=1)
    E = np.exp(A - rowmax[:, None]).astype(dtype)
    ahat = np.zeros((T, N), np.float64)
    logs = np.zeros((T, N), np.float64)
    m0 = x[0].max()
    e = np.exp(x[0] - m0).astype(dtype)          # alpha_0 = x_0
    C = m0
    ahat[0] = np.log(np.maximum(e.astype(np.float64), 1e-300))
    for t in range(1, T):
        z = x[t] + rowmax
        m = z.max()
        px = np.exp(z - m).astype(dtype)                       # off the chain
        s = (E @ e).astype(dtype)                              # the chain: N x N mat-vec in dtype
        v = (s * px).astype(dtype)
        vmax = v.max()
        e = (v / vmax).astype(dtype)
        C += m + np.log(float(vmax))                           # off the chain (vmax kept per frame, logged later)
        logs[t] = np.log(np.maximum(s.astype(np.float64), 1e-300))
        ahat[t] = np.log(np.maximum(e.astype(np.float64), 1e-300))
    return C + np.log(float(e.astype(np.float64).sum())), ahat, logs


def fac_forward_linear(x, trans, target, S, group=1):
    """x [T][N], trans [N][N], target [L] (first S entries valid).  Returns (loss, w1 [T][S]) -- w1_t[i] = the share of the STAY
    branch in alpha_t[i] (what fac_bwd_blk consumes).  fp64 mantissas; `group` adjacent positions share ONE power-of-two exponent
    (group = 1: an exponent per position; group = P: the positions a lane of the planned kernel holds; group = 0: ONE exponent for
    the whole frame -- which is NOT enough: positions far behind the lattice front fall 1e-308 below the frame's largest entry and
    vanish although the best path runs through them later; tests/test_asg_linear_domain.py keeps that failure on record)."""
    x = np.asarray(x, np.float64)
    A = np.asarray(trans, np.float64)
    y = np.asarray(target[:S], np.int64)
    T = x.shape[0]
    eS = np.exp(A[y, y])
    eP = np.zeros(S)
    eP[1:] = np.exp(A[y[1:], y[:-1]])
    G = S if group == 0 else group
    ng = (S + G - 1) // G
    gid = np.arange(S) // G
    NEGE = -(1 << 40)                                          # exponent of an all-zero group
    mant = np.zeros(S)
    expo = np.full(ng, NEGE, np.int64)                         # a[i] = mant[i] * 2 ** expo[gid[i]]
    mant[0] = 0.5
    expo[0] = 1
    w1 = np.zeros((T, S))
    base = x[0, y[0]]                                          # alpha_0[0] = x_0[y_0]; sum of the per-frame shifts m_t
    for t in range(1, T):
        xt = x[t, y]
        m = xt.max()
        ex = np.exp(xt - m)                                    # off the chain
        e_self = expo[gid]                                     # exponent of a_{t-1}[i]
        e_prev = np.concatenate(([NEGE], expo[gid[:-1]]))      # ... of a_{t-1}[i-1] (the neighbouring lane's at a group boundary)
        m_prev = np.concatenate(([0.0], mant[:-1]))
        # target exponent of every group: the largest exponent any of its positions sees
        e_in = np.maximum(e_self, np.where(m_prev > 0, e_prev, NEGE))
        e_grp = np.full(ng, NEGE, np.int64)
        np.maximum.at(e_grp, gid, e_in)
        E = e_grp[gid]
        stay = np.ldexp(mant * eS, np.clip(e_self - E, -2000, 0).astype(np.int64))
        adv = np.ldexp(m_prev * eP, np.clip(e_prev - E, -2000, 0).astype(np.int64))
        tot = stay + adv
        with np.errstate(invalid="ignore", divide="ignore"):
            w1[t] = np.where(tot > 0, stay / tot, 0.0)
        a = tot * ex
        # renormalise every group to its largest mantissa in [0.5, 1) -- exact
        gmax = np.zeros(ng)
        np.maximum.at(gmax, gid, a)
        k = np.where(gmax > 0, np.frexp(np.where(gmax > 0, gmax, 1.0))[1], 0).astype(np.int64)
        mant = np.ldexp(a, -k[gid])
        expo = np.where(gmax > 0, e_grp + k, NEGE)
        base += m
    last = S - 1
    if mant[last] <= 0:
        return -np.inf, w1
    return base + np.log(mant[last]) + float(expo[gid[last]]) * np.log(2.0), w1


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: models of the arithmetic the shipped N <= 31 kernels run (csrc/criterion_asg_small.hip), op for op where it
# matters for the numerics: the FCC scan with a LAGGED power-of-two scale (no maximum on the dependency chain), its backward
# scan in the same scaled domain (beta recursion, no logs), and the FAC scan with fp64 mantissas, one exponent per lane of P
# adjacent positions, renormalised every R frames with a decaying maximum-scan over the lane exponents.
def fcc_kernel_model(x, trans, dtype=np.float32, kclamp=64):
    """x [T][N], trans [N][N].  Returns (loss, u [T][N], q [T][N], ks [T]) as fcc_fwd_dpp computes them:
         u_0 = 2 ** (z_0 - max z_0),                       z_0 = x_0 log2(e)
         s_t = E u_{t-1},  u_t = s_t * q_t,                 q_t = 2 ** (zz_t - max zz_t) * 2 ** -k_t,  zz_t = x_t log2(e) + rowmax log2(e)
         k_{t+1} = clamp(exponent(sum_j u_{t-1}[j]) - k_t)  (the sum arrives as one more row of the mat-vec: E[31][j] = 1)
       so the magnitude of u_t is bounded by the growth of TWO frames and nothing but the mat-vec and one multiply is on the chain."""
    x = np.asarray(x, np.float64)
    A = np.asarray(trans, np.float64)
    T, N = x.shape
    L2E = 1.4426950408889634
    rowmax = A.max(axis=1)
    E = np.exp(A - rowmax[:, None]).astype(dtype)
    u = np.zeros((T, N), dtype)
    q = np.zeros((T, N), dtype)
    ks = np.zeros(T, np.int64)
    z0 = (x[0] * L2E).astype(dtype)
    C2 = float(z0.max())
    u[0] = np.exp2((z0 - z0.max()).astype(dtype)).astype(dtype)
    k = 0
    for t in range(1, T):
        zz = (x[t] * L2E + rowmax * L2E).astype(dtype)
        mz = zz.max()
        P = np.exp2((zz - mz).astype(dtype)).astype(dtype)
        s = (E @ u[t - 1]).astype(dtype)
        mass = dtype(u[t - 1].sum(dtype=dtype))          # what row 31 of the mat-vec delivers with s_t
        q[t] = np.ldexp(P, -k).astype(dtype)
        u[t] = (s * q[t]).astype(dtype)
        ks[t] = k
        C2 += float(mz) + k
        e = int(np.frexp(max(float(mass), 1e-45))[1]) - 1     # floor(log2(mass)) = the biased-exponent field - 127
        k = int(np.clip(e - k, -kclamp, kclamp))
    return C2 / L2E + float(np.log(u[T - 1].astype(np.float64).sum())), u, q, ks


def fcc_kernel_model_backward(u, q, trans, dtype=np.float32):
    """the backward scan of fcc_bwd_dpp on the forward's u, q: b_{T-1} = 1 / sum u_{T-1}; r_t = b_t q_t; b_{t-1} = E^T r_t;
    returns (dx [T][N] = u_t b_t (d loss / d x_t), dA [N][N] = E .* sum_t r_t u_{t-1}^T)"""
    A = np.asarray(trans, np.float64)
    T, N = u.shape
    E = np.exp(A - A.max(axis=1)[:, None]).astype(dtype)
    b = np.full(N, dtype(1.0) / dtype(u[T - 1].sum(dtype=dtype)), dtype)
    dx = np.zeros((T, N), np.float64)
    acc = np.zeros((N, N), np.float64)
    for t in range(T - 1, 0, -1):
        dx[t] = (u[t] * b).astype(dtype)
        r = (b * q[t]).astype(dtype)
        acc += np.outer(r.astype(np.float64), u[t - 1].astype(np.float64))
        b = (E.T @ r).astype(dtype)
    dx[0] = (u[0] * b).astype(dtype)
    return dx, E.astype(np.float64) * acc


def fac_kernel_model(x, trans, target, S, P=5, R=4, D=600):
    """x [T][N], trans [N][N], target[0..S).  Returns (loss, w1 [T][S]) as fac_fwd_lin computes them.
       h_t[i] = alpha_t[i] * exp(A[y_i][y_i]) (linear, frame-shifted): h_t[i] = c_t[y_i] * (h_{t-1}[i] + kappa[i] h_{t-1}[i-1]),
       c_t[n] = 2 ** (z_t[n] - max_n z_t[n]), z_t[n] = (x_t[n] + A[n][n]) log2(e) -- ONE row of N values per frame, computed with an
       integer / fraction split so that it cannot underflow -- kappa[i] = exp(A[y_i][y_{i-1}] - A[y_{i-1}][y_{i-1}]).
       Lane l holds positions l P .. l P + P - 1 as fp64 mantissas with ONE integer exponent e_l; the left neighbour's last
       position arrives scaled by 2 ** (e_{l-1} - e_l).  Every R frames -- and at once when some lane's largest mantissa has left
       [2 ** -200, 2 ** 300] -- the lanes renormalise:
         * positions that can no longer reach the end (i < S - (T - t): the reference never computes them, SURVEY App. B.1 `low`)
           are zeroed.  Without this a tight alignment (T ~ S) loses its answer: the lagging, useless positions outgrow the
           lattice front by thousands of bits, and the front IS the only path that finishes;
         * a lane with mass takes e_l = max(own, e_{l'} - D * (lanes with mass between l' and l)): what arrives from the left can
           be at most 2 ** D larger than what the lane holds (no overflow); what such a lane flushes lies 1000+ bits below a
           feasible position at most a few labels away, i.e. is negligible as long as one stay gains less than ~100 bits;
         * an empty lane copies the exponent of the nearest lane with mass on its left (the lattice front enters it unscaled)."""
    x = np.asarray(x, np.float64)
    A = np.asarray(trans, np.float64)
    y = np.asarray(target[:S], np.int64)
    T, N = x.shape
    L2E = 1.4426950408889634
    NL = (S + P - 1) // P
    SP = NL * P
    kappa = np.zeros(SP)
    kappa[1:S] = np.exp((A[y[1:], y[:-1]] - A[y[:-1], y[:-1]]).astype(np.float32)).astype(np.float32)
    yy = np.zeros(SP, np.int64)
    yy[:S] = y
    pos = np.arange(SP)
    valid = pos < S
    lane = pos // P
    NEGE = -(1 << 30)

    def crow(t):
        z = ((x[t] + np.diag(A)) * L2E).astype(np.float32)
        zm = z.max()
        zr = np.maximum((z - zm).astype(np.float32), -4000.0)
        zi = np.rint(zr)
        frac = np.exp2((zr - zi).astype(np.float32)).astype(np.float32)
        return np.ldexp(frac.astype(np.float64), zi.astype(np.int64)), float(zm)

    h = np.zeros(SP)
    e = np.zeros(NL, np.int64)
    d = np.zeros(NL, np.int64)
    c, zsum = crow(0)
    h[0] = c[yy[0]]
    w1 = np.zeros((T, S))

    def lane_max(h):
        return np.array([h[l * P:(l + 1) * P].max() for l in range(NL)])

    def renorm(h, e, t):
        h = np.where(pos >= S - (T - t), h, 0.0)                    # prune what cannot finish any more
        mx = lane_max(h)
        has = mx > 0
        k = np.where(has, np.frexp(np.where(has, mx, 1.0))[1], 0).astype(np.int64)
        cand = np.where(has, e + k, NEGE)
        cnt = np.cumsum(has)
        v = np.where(has, cand + D * cnt, NEGE)
        run = np.maximum.accumulate(v)
        en = np.where(run > NEGE, run - D * cnt, e)
        sh = np.clip(e - en, -2200, 2200)
        h = np.ldexp(h, sh[lane].astype(np.int64))
        d = np.zeros(NL, np.int64)
        d[1:] = np.clip(en[:-1] - en[1:], -2200, 2200)
        return h, en, d

    h, e, d = renorm(h, e, 0)
    for t in range(1, T):
        c, zm = crow(t)
        zsum += zm
        prev = np.concatenate(([0.0], h[:-1]))
        first = pos % P == 0
        prev = np.where(first, np.ldexp(prev, d[lane].astype(np.int64)), prev)
        tot = h + prev * kappa
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            w = (h * (1.0 / np.maximum(tot, 2.0 ** -1000))).astype(np.float32)
        w1[t] = np.where(valid, w, 0.0)[:S]
        h = np.where(valid, c[yy] * tot, 0.0)
        mxl = lane_max(h)
        if t % R == 0 or ((mxl > 0) & ((mxl < 2.0 ** -200) | (mxl > 2.0 ** 300))).any():
            h, e, d = renorm(h, e, t)
    last = S - 1
    if h[last] <= 0:
        return -np.inf, w1
    return (zsum + float(e[last // P])) / L2E + np.log(h[last]) - A[y[last], y[last]], w1


FAC_SAFE_BITS = 160.0   # kFacSafeBits of csrc/criterion_fac_lin.hpp


def fac_kernel_gain_bits(x, trans, target, S):
    """what fac_fwd_lin measures to decide whether its per-lane exponents are exact for an utterance: the largest per-frame
    spread of the label scores (x_t[n] + A[n][n]) log2 e plus the largest |log2 kappa| of the target; beyond FAC_SAFE_BITS the
    utterance is flagged and the log-domain kernel recomputes it"""
    x = np.asarray(x, np.float64)
    A = np.asarray(trans, np.float64)
    y = np.asarray(target[:S], np.int64)
    L2E = 1.4426950408889634
    z = ((x + np.diag(A)) * L2E).astype(np.float32)
    spread = float((z.max(axis=1) - z.min(axis=1)).max())
    kb = float(np.abs((A[y[1:], y[:-1]] - A[y[:-1], y[:-1]]).astype(np.float32)).max() * L2E) if S > 1 else 0.0
    return spread + kb
