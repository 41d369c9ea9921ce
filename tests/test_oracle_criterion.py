"""Pin the CPU oracle for the sequence criteria.

The reference holds no golden vectors for FAC/FCC/Viterbi/CTC ("parity
unpinned"), so the oracle is pinned by brute-force path enumeration, analytic
identities, fp64 finite differences and torch's CPU CTC.
"""
import itertools

import numpy as np
import pytest
import torch


def lse(v):
    v = np.asarray(v, np.float64)
    m = v.max()
    return m + np.log(np.exp(v - m).sum())


def brute_fcc(x, A):
    T, N = x.shape
    scores = []
    for path in itertools.product(range(N), repeat=T):
        s = x[0, path[0]]
        for t in range(1, T):
            s += x[t, path[t]] + A[path[t], path[t - 1]]
        scores.append(s)
    return lse(scores)


def brute_fac(x, A, y):
    """all monotone alignments of y (len L) to T frames, each label >= 1 frame"""
    T, N = x.shape
    L = len(y)
    scores = []
    for cuts in itertools.combinations(range(1, T), L - 1):
        bounds = (0,) + cuts + (T,)
        pos = []
        for i in range(L):
            pos += [i] * (bounds[i + 1] - bounds[i])
        s = x[0, y[pos[0]]]
        for t in range(1, T):
            s += x[t, y[pos[t]]] + A[y[pos[t]], y[pos[t - 1]]]
        scores.append(s)
    return lse(scores)


def brute_viterbi(x, A):
    T, N = x.shape
    best, arg = -np.inf, None
    for path in itertools.product(range(N), repeat=T):
        s = x[0, path[0]]
        for t in range(1, T):
            s += x[t, path[t]] + A[path[t], path[t - 1]]
        if s > best:
            best, arg = s, path
    return best, arg


@pytest.mark.parametrize("seed", range(6))
def test_fcc_bruteforce(oracle, seed):
    rng = np.random.default_rng(seed)
    T, N = int(rng.integers(2, 7)), int(rng.integers(2, 5))
    x = rng.normal(size=(1, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    got = oracle.FCC(x, A, [1]).forward()[0]
    assert abs(got - brute_fcc(x[0].astype(np.float64), A.astype(np.float64))) < 1e-10


@pytest.mark.parametrize("seed", range(8))
def test_fac_bruteforce(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    T, N = int(rng.integers(3, 9)), int(rng.integers(2, 5))
    L = int(rng.integers(1, T + 1))
    y = rng.integers(0, N, size=L)
    x = rng.normal(size=(1, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = np.full((1, L + 2), -1, np.int32)
    tgt[0, :L] = y
    got = oracle.FAC(x, A, tgt).forward()[0]
    assert abs(got - brute_fac(x[0].astype(np.float64), A.astype(np.float64), list(y))) < 1e-10


@pytest.mark.parametrize("seed", range(6))
def test_viterbi_bruteforce(oracle, seed):
    rng = np.random.default_rng(200 + seed)
    T, N = int(rng.integers(2, 7)), int(rng.integers(2, 5))
    x = rng.normal(size=(1, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    path = oracle.viterbi(x, A)[0]
    best, arg = brute_viterbi(x[0].astype(np.float64), A.astype(np.float64))
    assert tuple(path) == arg


def test_viterbi_tie_break_first_max(oracle):
    # all-equal scores: first index must win everywhere (strict '>' scan)
    x = np.zeros((2, 5, 4), np.float32)
    A = np.zeros((4, 4), np.float32)
    assert (oracle.viterbi(x, A) == 0).all()
    assert (oracle.ctc_viterbi(x) == 0).all()


def test_identities_b6(oracle):
    rng = np.random.default_rng(7)
    B, T, N = 3, 9, 5
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    # (i) A = 0 => FCC = sum_t LSE_n x
    fcc = oracle.FCC(x, np.zeros((N, N), np.float32), [1] * B)
    want = np.array([sum(lse(x[b, t]) for t in range(T)) for b in range(B)])
    assert np.allclose(fcc.forward(), want, atol=1e-10)
    A = rng.normal(size=(N, N)).astype(np.float32)
    # (ii) T == L => FAC = sum x[t][y_t] + sum A[y_t][y_t-1]
    y = rng.integers(0, N, size=(B, T)).astype(np.int32)
    fac = oracle.FAC(x, A, y)
    want = np.array([sum(float(x[b, t, y[b, t]]) for t in range(T)) +
                     sum(float(A[y[b, t], y[b, t - 1]]) for t in range(1, T)) for b in range(B)])
    assert np.allclose(fac.forward(), want, atol=1e-10)
    # (iii) L == 1
    y1 = np.full((B, 3), -1, np.int32)
    y1[:, 0] = [0, 2, 4]
    want = np.array([x[b, :, y1[b, 0]].astype(np.float64).sum() + (T - 1) * float(A[y1[b, 0], y1[b, 0]])
                     for b in range(B)])
    assert np.allclose(oracle.FAC(x, A, y1).forward(), want, atol=1e-10)
    # (iv),(v) gradient mass
    yl = np.full((B, 6), -1, np.int32)
    yl[0, :4] = [1, 2, 2, 0]; yl[1, :2] = [3, 3]; yl[2, :6] = [0, 1, 0, 1, 4, 4]
    for crit in (oracle.FCC(x, A, [4, 2, 6]), oracle.FAC(x, A, yl)):
        crit.forward()
        dx, dA = crit.backward()
        assert np.allclose(dx.sum(-1), 1.0, atol=1e-10)
        assert abs(dA.sum() - B * (T - 1)) < 1e-9


def _fd(f, x, eps=1e-3):
    g = np.zeros(x.shape, np.float64)
    it = np.nditer(x, flags=["multi_index"])
    while not it.finished:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; fp = f()
        x[i] = old - eps; fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
        it.iternext()
    return g


def test_asg_finite_differences(oracle):
    rng = np.random.default_rng(11)
    B, T, N, L = 2, 6, 4, 3
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32) * 0.5
    y = np.array([[1, 2, -1], [0, 0, 3]], np.int32)
    w = np.array([0.7, -1.3])
    mode = oracle.SCALE_TARGET_SZ_SQRT

    def f():
        return float((oracle.asg(x, A, y, mode)[0] * w).sum())

    _, dx, dA = oracle.asg(x, A, y, mode, grad=w)
    # float32 storage => central differences with eps 1e-2..1e-3; tolerance accordingly
    assert np.allclose(_fd(f, x, 1e-2), dx, atol=2e-3)
    assert np.allclose(_fd(f, A, 1e-2), dA, atol=2e-3)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_scale_modes(oracle, mode):
    rng = np.random.default_rng(3)
    B, T, N = 2, 16, 4
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = np.zeros((N, N), np.float32)
    y = np.array([[1, 2, 0, 1, -1], [3, -1, -1, -1, -1]], np.int32)
    base = oracle.FAC(x, A, y, scale_mode=0).forward()
    got = oracle.FAC(x, A, y, scale_mode=mode).forward()
    Ls = np.array([4, 1])
    s = {0: np.ones(2), 1: np.full(2, 1 / T), 2: np.full(2, np.sqrt(1 / T)), 3: 1 / Ls,
         4: np.sqrt(1 / Ls)}[mode]
    assert np.allclose(got, base * s, rtol=1e-12)


@pytest.mark.parametrize("seed", range(5))
def test_ctc_vs_torch(oracle, seed):
    rng = np.random.default_rng(300 + seed)
    B, T, N, L = 4, int(rng.integers(8, 30)), int(rng.integers(3, 12)), 6
    x = (rng.normal(size=(B, T, N)) * 2).astype(np.float32)
    tgt = np.full((B, L), -1, np.int32)
    lens = []
    for b in range(B):
        l = int(rng.integers(0, L + 1)) if b else L
        tgt[b, :l] = rng.integers(0, N - 1, size=l)
        lens.append(l)
    if seed == 0:
        tgt[0, :L] = [1, 1, 1, 0, 0, 1]  # repeats
    ctc = oracle.CTC(x, tgt)
    loss = ctc.forward()
    w = rng.normal(size=B)
    dx = ctc.backward(w)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(xt, -1).transpose(0, 1)  # T,B,N
    tl = torch.tensor(lens)
    tt = torch.tensor(np.where(tgt < 0, 0, tgt), dtype=torch.long)
    ref = torch.nn.functional.ctc_loss(lp, tt, torch.full((B,), T), tl, blank=N - 1, reduction="none")
    assert np.allclose(loss, ref.detach().numpy(), atol=1e-8)
    (ref * torch.tensor(w)).sum().backward()
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-8)


def test_ctc_long_transcription_vs_torch(oracle):
    """the oracle at transcriptions longer than 255 labels (the shapes tests/test_gpu_criterion.py::test_ctc_long_transcriptions
    holds the HIP scan to) against torch's ctc_loss, loss and gradient"""
    rng = np.random.default_rng(77)
    B, T, N, L = 2, 700, 40, 300
    x = (rng.normal(size=(B, T, N)) * 2).astype(np.float32)
    tgt = rng.integers(0, N - 1, size=(B, L)).astype(np.int32)
    tgt[0, 10:14] = [1, 1, 1, 0]
    ctc = oracle.CTC(x, tgt)
    loss = ctc.forward()
    w = rng.normal(size=B)
    dx = ctc.backward(w)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(xt, -1).transpose(0, 1)
    ref = torch.nn.functional.ctc_loss(lp, torch.tensor(tgt, dtype=torch.long), torch.full((B,), T), torch.full((B,), L),
                                       blank=N - 1, reduction="none")
    assert np.allclose(loss, ref.detach().numpy(), rtol=1e-10, atol=1e-7)
    (ref * torch.tensor(w)).sum().backward()
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-8)


def test_ctc_target_truncation(oracle):
    # L + repeats > T: target size is reduced to what fits (App. B.0)
    tgt = np.array([[1, 1, 1, 1, -1]], np.int32)
    assert oracle.batch_ctc_target_size(tgt, 5)[0] == 2  # R=3: min(4+3,5)-3
    assert oracle.batch_ctc_target_size(tgt, 7)[0] == 4
    assert oracle.batch_target_size(tgt, 3)[0] == 3


def test_ctc_identity_T_equals_L(oracle):
    rng = np.random.default_rng(5)
    T, N = 6, 7
    x = rng.normal(size=(1, T, N)).astype(np.float32)
    y = np.array([[0, 1, 2, 3, 4, 5]], np.int32)
    lp = x[0].astype(np.float64) - np.array([lse(r) for r in x[0]])[:, None]
    want = -sum(lp[t, y[0, t]] for t in range(T))
    assert abs(oracle.CTC(x, y).forward()[0] - want) < 1e-10


def test_fac_viterbi_matches_enumeration(oracle):
    rng = np.random.default_rng(9)
    T, N, L = 7, 4, 3
    x = rng.normal(size=(1, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    y = [2, 0, 3]
    tgt = np.array([y + [-1]], np.int32)
    got = oracle.FAC(x, A, tgt).viterbi()[0]
    best, arg = -np.inf, None
    for cuts in itertools.combinations(range(1, T), L - 1):
        bounds = (0,) + cuts + (T,)
        pos = []
        for i in range(L):
            pos += [i] * (bounds[i + 1] - bounds[i])
        s = float(x[0, 0, y[pos[0]]])
        for t in range(1, T):
            s += float(x[0, t, y[pos[t]]]) + float(A[y[pos[t]], y[pos[t - 1]]])
        if s > best:
            best, arg = s, [y[p] for p in pos]
    assert list(got) == arg


def test_viterbi_score_le_fcc(oracle):
    rng = np.random.default_rng(13)
    B, T, N = 3, 12, 6
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    p = oracle.viterbi(x, A)
    fcc = oracle.FCC(x, A, [1] * B).forward()
    for b in range(B):
        s = float(x[b, 0, p[b, 0]]) + sum(float(x[b, t, p[b, t]]) + float(A[p[b, t], p[b, t - 1]])
                                          for t in range(1, T))
        assert s <= fcc[b] + 1e-9
    big = oracle.FCC(x * 50, A * 50, [1] * B).forward()
    p2 = oracle.viterbi(x * 50, A * 50)
    for b in range(B):
        s = 50 * (float(x[b, 0, p2[b, 0]]) + sum(float(x[b, t, p2[b, t]]) + float(A[p2[b, t], p2[b, t - 1]])
                                                 for t in range(1, T)))
        assert abs(s - big[b]) < 1e-3 * abs(big[b])


def test_linear_target_properties(oracle):
    """getLinearTarget restatement: every label appears, in order, over contiguous frame runs whose lengths differ by
    at most one; L = T is the identity; an empty / too long target gives a row of -1"""
    rng = np.random.default_rng(5)
    tgt = np.full((5, 9), -1, np.int32)
    lens = [1, 4, 9, 0, 7]
    for b, l in enumerate(lens):
        tgt[b, :l] = rng.integers(0, 6, size=l)
    T = 8
    lin = oracle.linear_target(tgt, T)
    assert lin.shape == (5, T)
    assert (lin[0] == tgt[0, 0]).all()
    assert (lin[2] == -1).all() and (lin[3] == -1).all()       # L = 9 > T; L = 0
    for b in (1, 4):
        l = lens[b]
        idx = (np.arange(T) * l) // T
        assert (lin[b] == tgt[b, idx]).all()
        assert sorted(set(idx)) == list(range(l))                 # every label is used
        runs = np.bincount(idx)
        assert runs.max() - runs.min() <= 1
    sq = oracle.linear_target(tgt[4:5, :7], 7)
    assert (sq[0] == tgt[4, :7]).all()


def test_linseg_is_single_path_asg(oracle):
    """LinSeg = FCC - (score of the one linear alignment): the oracle's FAC on a length-T target equals the direct
    path score, and its gradient is the path indicator"""
    rng = np.random.default_rng(6)
    B, T, N, L = 3, 7, 4, 5
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = np.array([[1, 2, 3, -1, -1], [0, 0, 1, 2, 3], [2, -1, -1, -1, -1]], np.int32)
    lin = oracle.linear_target(tgt, T)
    loss, dx, dA = oracle.linseg(x, A, tgt)
    fcc = oracle.FCC(x, A, np.full(B, T, np.int32))
    lf = fcc.forward()
    dxf, dAf = fcc.backward()
    for b in range(B):
        s = x[b, 0, lin[b, 0]].astype(np.float64)
        ind = np.zeros((T, N))
        ind[0, lin[b, 0]] = 1
        cnt = np.zeros((N, N))
        for t in range(1, T):
            s += float(x[b, t, lin[b, t]]) + float(A[lin[b, t], lin[b, t - 1]])
            ind[t, lin[b, t]] = 1
            cnt[lin[b, t], lin[b, t - 1]] += 1
        assert abs(loss[b] - (lf[b] - s)) < 1e-9
        assert np.abs(dx[b] - (dxf[b] - ind)).max() < 1e-9
    assert np.isfinite(dA).all()


def test_linseg_finite_differences(oracle):
    """the LinSeg restatement's gradients (emissions and transitions) against central differences of its own loss"""
    rng = np.random.default_rng(12)
    B, T, N, L = 2, 6, 3, 4
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.5).astype(np.float32)
    tgt = np.array([[0, 2, 1, -1], [1, 1, 0, 2]], np.int32)
    g = np.array([0.7, 1.3])
    loss, dx, dA = oracle.linseg(x, A, tgt, grad=g)
    f = lambda xx, AA: float((oracle.linseg(xx, AA, tgt)[0] * g).sum())
    eps = 1e-2
    for idx in [(0, 0, 0), (0, 3, 2), (1, 5, 1), (1, 2, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps; xm[idx] -= eps
        assert abs((f(xp, A) - f(xm, A)) / (2 * eps) - dx[idx]) < 5e-3
    for idx in [(0, 0), (2, 1), (1, 2)]:
        Ap, Am = A.copy(), A.copy()
        Ap[idx] += eps; Am[idx] -= eps
        assert abs((f(x, Ap) - f(x, Am)) / (2 * eps) - dA[idx]) < 5e-3


def test_fcc_fac_thread_layouts_agree(oracle):
    """few utterances x many labels puts the OpenMP threads on the label loop instead of the batch loop
    (the N = 9998 parity checks); both layouts are the same arithmetic"""
    rng = np.random.default_rng(5)
    B, T, N, L = 2, 5, 600, 4
    x = rng.normal(size=(B, T, N)).astype(np.float32)
    A = rng.normal(size=(N, N)).astype(np.float32)
    tgt = rng.integers(0, N, size=(B, L)).astype(np.int32)
    ts = oracle.batch_target_size(tgt, T)
    w = rng.normal(size=B)
    n0 = oracle.num_threads()
    res = []
    try:
        for nthr in (1, 4):   # 1: batch loop (serial); 4 > B: label loop
            oracle.set_num_threads(nthr)
            o = oracle.FCC(x, A, ts, 4)
            l = o.forward()
            dx, dA = o.backward(w)
            f = oracle.FAC(x, A, tgt, scale_mode=4)
            fl = f.forward()
            fdx, fdA = f.backward(w)
            res.append((l, dx, dA, fl, fdx, fdA))
    finally:
        oracle.set_num_threads(n0)
    for a, b in zip(*res):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())


def test_criterion_handover_vector(oracle):
    """tests/golden/criterion_handover.json (generator alongside): the oracle reproduces it, ASG = FCC - FAC, and its CTC
    numbers agree with torch's ctc_loss (blank last) -- the one second implementation available here"""
    import json
    import os
    import torch
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "criterion_handover.json")))
    B, T, N, L = g["B"], g["T"], g["N"], g["L"]
    em = np.array(g["emissions"], np.float32).reshape(B, T, N)
    A = np.array(g["transitions"], np.float32).reshape(N, N)
    tgt = np.array(g["target"], np.int32).reshape(B, L)
    loss, dx, dA = oracle.asg(em, A, tgt)
    assert np.abs(loss - np.array(g["asg_loss"])).max() < 1e-9
    assert np.abs(np.array(g["fcc"]) - np.array(g["fac"]) - np.array(g["asg_loss"])).max() < 1e-9
    assert (oracle.viterbi(em, A).reshape(-1) == np.array(g["viterbi_path"])).all()
    assert np.abs(np.asarray(dA).reshape(-1) - np.array(g["asg_grad_transitions"])).max() < 1e-9
    ct = np.array(g["ctc_target"], np.int32).reshape(B, L)
    lens = (ct >= 0).sum(1)
    lp = torch.log_softmax(torch.tensor(em, dtype=torch.float64), -1).permute(1, 0, 2)
    want = torch.nn.functional.ctc_loss(lp, torch.tensor(np.where(ct >= 0, ct, 0)), torch.full((B,), T), torch.tensor(lens),
                                        blank=N - 1, reduction="none")
    assert np.abs(want.numpy() - np.array(g["ctc_loss"])).max() < 1e-9
    assert np.abs(oracle.CTC(em, ct).forward() - np.array(g["ctc_loss"])).max() < 1e-9


def _torch_asg(x, A, y):
    """independent restatement for the test below: the two recursions as torch float64 logsumexp loops (vectorised over states /
    positions), gradients by autograd -- no code shared with oracle/criterion_oracle.c"""
    T, N = x.shape
    S = len(y)
    alpha = x[0]
    for t in range(1, T):                                  # FCC: alpha_t[j] = x_t[j] + lse_i(alpha_{t-1}[i] + A[j][i])
        alpha = x[t] + torch.logsumexp(alpha.unsqueeze(0) + A, dim=1)
    fcc = torch.logsumexp(alpha, dim=0)
    yt = torch.tensor(y, dtype=torch.long)
    self_t = A[yt, yt]
    prev_t = torch.cat([torch.zeros(1, dtype=torch.float64), A[yt[1:], yt[:-1]]])
    neg = torch.full((1,), -1e30, dtype=torch.float64)     # "unreachable": finite, so that autograd of lse(-inf, -inf) is 0, not NaN
    a = torch.cat([x[0, yt[:1]], neg.expand(S - 1)]) if S > 1 else x[0, yt[:1]]
    for t in range(1, T):                                  # FAC: stay or advance from position i - 1
        stay = a + self_t
        adv = torch.cat([neg, a[:-1] + prev_t[1:]]) if S > 1 else neg
        a = x[t, yt] + torch.logsumexp(torch.stack([stay, adv]), dim=0)
    return fcc, a[S - 1]


@pytest.mark.parametrize("T,N,L,S,scale", [(60, 8, 12, 9, 1.0), (400, 30, 120, 77, 1.0), (2000, 30, 300, 300, 1.0), (700, 30, 300, 41, 6.0)])
def test_fcc_fac_against_torch_autograd_at_recipe_scale(oracle, T, N, L, S, scale):
    """a second, independent implementation at the conv_glu recipe's sizes (T = 2000 frames, N = 30 tokens, 300 labels), where
    brute force cannot go: losses to 1e-10 relative, the oracle's hand-written backward passes (input and transition gradients
    of FCC and FAC separately) against autograd to 1e-8 of the largest entry"""
    rng = np.random.default_rng(T + S)
    x = (rng.normal(size=(1, T, N)) * scale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.5 + np.eye(N)).astype(np.float32)
    tgt = np.full((1, L), -1, np.int32)
    tgt[0, :S] = rng.integers(0, N, S)
    xt = torch.tensor(x[0], dtype=torch.float64, requires_grad=True)
    At = torch.tensor(A, dtype=torch.float64, requires_grad=True)
    fcc_t, fac_t = _torch_asg(xt, At, tgt[0, :S].tolist())
    fac = oracle.FAC(x, A, tgt)
    fcc = oracle.FCC(x, A, fac.ts)
    lf, la = fcc.forward()[0], fac.forward()[0]
    assert abs(lf - fcc_t.item()) < 1e-10 * abs(lf) and abs(la - fac_t.item()) < 1e-10 * max(1.0, abs(la))
    for loss_t, o in ((fcc_t, fcc), (fac_t, fac)):
        gx, gA = torch.autograd.grad(loss_t, (xt, At), retain_graph=True)
        dx, dA = o.backward()
        assert np.abs(dx[0] - gx.numpy()).max() < 1e-8 * max(1.0, np.abs(gx.numpy()).max())
        assert np.abs(dA - gA.numpy()).max() < 1e-8 * max(1.0, np.abs(gA.numpy()).max())


@pytest.mark.parametrize("T,N,scale", [(2000, 30, 1.0), (300, 200, 3.0), (64, 9998, 1.0)])
def test_viterbi_against_an_independent_dp_at_recipe_scale(oracle, T, N, scale):
    """ViterbiPath at sizes brute force cannot reach (the conv_glu recipe's T = 2000 x N = 30, and N = 9998 word pieces): a
    vectorised numpy max-product DP in float32 -- the oracle's arithmetic type -- with argmax taking the FIRST maximum, as the
    reference's strict `>` scan does; the paths must be identical"""
    rng = np.random.default_rng(T + N)
    x = (rng.normal(size=(1, T, N)) * scale).astype(np.float32)
    A = (rng.normal(size=(N, N)) * 0.5).astype(np.float32)
    score = x[0, 0].copy()
    back = np.zeros((T, N), np.int32)
    for t in range(1, T):
        cand = score[None, :] + A                      # cand[to][from], float32
        back[t] = cand.argmax(axis=1)                  # first maximum
        score = (cand.max(axis=1) + x[0, t]).astype(np.float32)
    path = np.zeros(T, np.int32)
    path[T - 1] = int(score.argmax())
    for t in range(T - 1, 0, -1):
        path[t - 1] = back[t, path[t]]
    assert (oracle.viterbi(x, A)[0] == path).all()
