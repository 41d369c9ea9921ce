"""Randomised range check of the sequence criteria against the fp64 oracle: FCC / FAC / ASG (loss, dx, dA) and CTC over label-set sizes on
both sides of every kernel switch (N <= 31 / 64 / large), lattice widths, emission and transition magnitudes far outside the recipes'.
Prints one line per case, `BAD` where the device result is non-finite while the oracle is finite or the error passes 1e-3.
   python tools/exp/criterion_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import pyoracle as O
from wav2letter_amd import ASGLoss, CTCLoss, ForceAlignmentCriterion, FullConnectionCriterion

def rel(got, want):
    want = np.asarray(want, np.float64); got = np.asarray(got, np.float64)
    if not np.isfinite(got).all():
        return np.inf
    # (gradients are posteriors times weight and scale: an utterance whose loss is ~0 has gradients of 1e-12 that are all rounding;
    #  relative to the largest entry, but never to less than 1e-3)
    return float(np.abs(got - want).max() / max(1e-3, np.abs(want).max()))



def run(cases, seed, x_scales=(0.1, 1.0, 5.0, 20.0, 50.0), a_scales=(0.0, 0.3, 2.0, 8.0, 20.0, 40.0), verbose=True,
        n_choices=(3, 5, 16, 29, 30, 31, 32, 33, 40, 64, 65, 100, 1000)):
    """-> the lines of the cases whose device result is non-finite where the oracle's is finite, or off by more than 1e-3"""
    rng = np.random.default_rng(seed)
    bad = 0
    bad_lines = []
    for c in range(cases):
        N = int(rng.choice(n_choices))
        T = int(rng.choice([1, 2, 17, 50, 300, 1000, 2000]))
        if N >= 500:
            T = min(T, 300)   # (the fp64 oracle is N^2 T per utterance on the host)
        B = int(rng.integers(1, 4))
        Lmax = int(rng.choice([1, 2, 7, 64, 65, 128, 200, 300]))
        xs = float(rng.choice(x_scales))
        as_ = float(rng.choice(a_scales))
        diag = float(rng.choice([0.0, 4.0]))
        mode = int(rng.choice([0, 1, 2, 3, 4]))
        x = (rng.normal(size=(B, T, N)) * xs).astype(np.float32)
        A = (rng.normal(size=(N, N)) * as_ + np.eye(N) * diag).astype(np.float32)
        tgt = np.full((B, Lmax), -1, np.int32)
        for b in range(B):
            l = int(rng.integers(1, min(Lmax, T) + 1))
            y = rng.integers(0, N, size=l)
            tgt[b, :l] = y
        ts = O.batch_target_size(tgt, T)
        w = rng.uniform(0.5, 1.5, size=B)
        line = f"case {c:3d} B={B} T={T:4d} N={N:3d} L<={Lmax:3d} x*{xs:<4} A*{as_:<4} diag {diag} mode {mode}:"
        class _AsgOracle:   # FCC - FAC on the same transitions (oracle/pyoracle.py::asg)
            def forward(self):
                self.r = O.asg(x, A, tgt, mode, w)
                return self.r[0]
            def backward(self, _w):
                return self.r[1], self.r[2]
        for name, cls, orc in (("FCC", FullConnectionCriterion, lambda: O.FCC(x, A, ts, mode)), ("FAC", ForceAlignmentCriterion, lambda: O.FAC(x, A, tgt, scale_mode=mode)),
                               ("ASG", lambda n_, m_: ASGLoss(n_, m_, 0.0), _AsgOracle)):   # ASGLoss: w2l_asg_forward / w2l_asg_backward (N <= 31: the fused sequence)
            crit = cls(N, mode).cuda()
            crit.transitions.data = torch.from_numpy(A).cuda()
            xt = torch.from_numpy(x).cuda().requires_grad_(True)
            loss = crit(xt, torch.from_numpy(tgt).cuda())
            (loss * torch.from_numpy(w.astype(np.float32)).cuda()).sum().backward()
            o = orc()
            ol = o.forward()
            odx, odA = o.backward(w)
            if not np.isfinite(ol).all():
                line += f" {name} oracle-nonfinite"
                continue
            el = float(np.abs(loss.detach().cpu().numpy().astype(np.float64) - ol).max() / max(1.0, np.abs(ol).max())) if torch.isfinite(loss).all() else np.inf
            ex, ea = rel(xt.grad.cpu().numpy(), odx), rel(crit.transitions.grad.cpu().numpy(), odA)
            flag = "" if max(el, ex, ea) < 1e-3 else " BAD"
            bad += bool(flag)
            line += f" {name} {el:.1e}/{ex:.1e}/{ea:.1e}{flag}"
        # Viterbi paths: bit-exact (the free path of ASGLoss::viterbiPath and the forced alignment)
        asg = ASGLoss(N, mode, 0.0).cuda()
        asg.transitions.data = torch.from_numpy(A).cuda()
        vp = asg.viterbiPath(torch.from_numpy(x).cuda()).cpu().numpy()
        okv = bool((vp == O.viterbi(x, A)).all())
        fac = ForceAlignmentCriterion(N, mode).cuda()
        fac.transitions.data = torch.from_numpy(A).cuda()
        fp = fac.viterbiPath(torch.from_numpy(x).cuda(), torch.from_numpy(tgt).cuda()).cpu().numpy()
        okf_ = bool((fp == O.FAC(x, A, tgt, scale_mode=mode).viterbi()).all())
        if not (okv and okf_):
            bad += 1
            line += f" VITERBI free {okv} forced {okf_} BAD"
        # CTC: N includes the blank (last index); targets without the blank
        if N >= 2 and T >= 1:
            tg = np.where(tgt >= 0, np.minimum(tgt, N - 2), -1).astype(np.int32)
            crit = CTCLoss(mode)
            xt = torch.from_numpy(x).cuda().requires_grad_(True)
            loss = crit(xt, torch.from_numpy(tg).cuda())
            fin = torch.isfinite(loss)
            o = O.CTC(x, tg, scale_mode=mode)
            ol = o.forward()
            okf = np.isfinite(ol)
            if okf.any():
                wz = np.where(okf, w, 0.0)
                (torch.where(fin, loss, torch.zeros_like(loss)) * torch.from_numpy(wz.astype(np.float32)).cuda()).sum().backward()
                got = loss.detach().cpu().numpy().astype(np.float64)
                el = float(np.abs(got[okf] - ol[okf]).max() / max(1.0, np.abs(ol[okf]).max())) if np.isfinite(got[okf]).all() else np.inf
                same_inf = bool((np.isfinite(got) == okf).all())
                ex = rel(xt.grad.cpu().numpy(), o.backward(wz))
                flag = "" if max(el, ex) < 1e-3 and same_inf else " BAD"
                bad += bool(flag)
                line += f" CTC {el:.1e}/{ex:.1e}{'' if same_inf else ' inf-mismatch'}{flag}"
            else:
                line += " CTC all-infeasible"
        if verbose:
            print(line, flush=True)
        if 'BAD' in line:
            bad_lines.append(line)
    return bad_lines


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    lines = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    print(f"{len(lines)} BAD of {n} cases")
