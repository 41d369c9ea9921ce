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
