"""TEST / BASELINE INFRASTRUCTURE (see oracle/criterion_oracle.c header): only tests/ and the
cpu_baseline leg of bench.py may import this.

torch-CPU (oneDNN / MKL) interpreter of the arch tokens of the TDS-CTC recipe family, forward and
backward through torch autograd.  This is the PROXY of the reference's CPU path that BASELINE.md sec. 3
item 2 and SURVEY.md 8(d) prescribe: Flashlight's CPU backend sits on the same oneDNN / MKL libraries
(recorded by the reference's notebook build log, recipes/mling_pl/mling_model.ipynb:916-919, :1075), the
reference itself (ArrayFire + Flashlight) cannot be built here.  Same layouts and parameter order as
oracle/refnet.RefNet ([B][C][H][T] activations == ArrayFire dims (T,H,C,B)); checked against it in
tests/test_oracle_nn.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pyoracle as O


class TorchNet:
    def __init__(self, arch_text, nfeat, nlabel):
        self.lines = []
        for raw in arch_text.splitlines():
            l = raw.strip()
            if not l or l.startswith("#"):
                continue
            self.lines.append(l.replace("NFEAT", str(nfeat)).replace("NLABEL", str(nlabel)).split())

    @staticmethod
    def _ln(a, mode, gb):
        if mode == "all":
            y = F.layer_norm(a, a.shape[1:], eps=1e-5)
        else:  # per frame: over (C, H)
            y = F.layer_norm(a.permute(0, 3, 1, 2), a.shape[1:3], eps=1e-5).permute(0, 2, 3, 1)
        return y * gb[0] + gb[1]

    @staticmethod
    def _conv(a, w, b, stride, pl, pr):
        # a [B][C][H][T], w [cout][cin][kw]: kw x 1 convolution over time
        return F.conv2d(F.pad(a, (pl, pr)), w[:, :, None, :], b, stride=(1, stride))

    def forward(self, x, params):
        """x: torch [B][1][NFEAT][T]; params: list of torch tensors in refnet order -> emissions [B][T'][N]"""
        a = x
        pi = 0
        for t in self.lines:
            k = t[0]
            if k in ("SAUG",):
                continue
            if k == "PD":   # zero padding of the time axis ahead of an unpadded convolution (streaming arch)
                assert float(t[1]) == 0.0 and all(int(v) == 0 for v in t[4:]), t
                a = F.pad(a, (int(t[2]), int(t[3])))
                continue
            if k == "V":
                dims = [int(v) for v in t[1:5]]
                cur = list(a.shape[::-1])
                for i in range(4):
                    if dims[i] == 0:
                        dims[i] = cur[i]
                if -1 in dims:
                    j = dims.index(-1)
                    dims[j] = int(a.numel() // np.prod([d for d in dims if d != -1]))
                a = a.contiguous().reshape(dims[::-1])
            elif k == "RO":
                p = [int(v) for v in t[1:5]]
                axes = [0] * 4
                for i in range(4):
                    axes[3 - i] = 3 - p[i]
                a = a.permute(axes)
            elif k == "C2":
                kw, stride = int(t[3]), int(t[5])
                pad = int(t[7]) if len(t) > 7 else 0
                if pad == -1:
                    pad = O.same_pad(a.shape[3], kw, stride)
                kh = int(t[4])
                if kh > 1:   # [cout][cin][kh][kw] kernel, SAME on the mel axis
                    ph = (kh - 1) // 2
                    a = F.conv2d(F.pad(a, (pad, pad, ph, ph)), params[pi], params[pi + 1], stride=(1, stride))
                else:
                    a = self._conv(a, params[pi], params[pi + 1], stride, pad, pad)
                pi += 2
            elif k == "L":
                nin = int(t[1])
                assert a.shape[3] == nin
                a = F.linear(a, params[pi].t(), params[pi + 1])  # memory [in][out]
                pi += 2
            elif k == "R":
                a = F.relu(a)
            elif k == "DO":
                assert float(t[1]) == 0.0
            elif k == "LN":
                axes = sorted(int(v) for v in t[1:])
                a = self._ln(a, "all" if axes == [0, 1, 2] else "frame", params[pi])
                pi += 1
            elif k == "TDS":
                c, kw, h = int(t[1]), int(t[2]), int(t[3])
                wc, bc, gb1, w1, b1, w2, b2, gb2 = params[pi:pi + 8]
                pi += 8
                rpad = int(t[6]) if len(t) > 6 else -1
                if rpad < 0:
                    pl = pr = O.same_pad(a.shape[3], kw, 1)
                else:
                    pr, pl = rpad, kw - 1 - rpad
                mode = "frame" if (len(t) > 7 and int(t[7]) == 0) else "all"
                B, Cc, H, T = a.shape
                y = self._ln(F.relu(self._conv(a, wc, bc, 1, pl, pr)) + a, mode, gb1)
                z = y.permute(0, 3, 2, 1).reshape(B * T, H * Cc)            # feature f = h*C + c
                v = F.linear(F.relu(F.linear(z, w1.t(), b1)), w2.t(), b2)
                a = self._ln(v.reshape(B, T, H, Cc).permute(0, 3, 2, 1) + y, mode, gb2)
            else:
                raise ValueError(f"torch-CPU proxy: token {k} not in the TDS-CTC family")
        assert a.shape[0] == 1, a.shape
        return a[0]


def tds_ctc_step_seconds(arch_text, nfeat, nlabel, B, T, L=20, warmup=2, runs=5, seed=0):
    """median wall time of one training step (network forward + CTC + backward; no optimizer) of the TDS-CTC
    recipe on the host: torch-CPU for the network with every core, the OpenMP oracle for the criterion"""
    import time
    from oracle import refnet
    arch = "\n".join(l for l in arch_text.splitlines() if not l.startswith("SAUG"))
    arch = "\n".join((" ".join(f[:4] + ["0.0"] + f[5:]) if f and f[0] == "TDS" else " ".join(f))
                     for f in (l.split() for l in arch.splitlines())) + "\n"
    rng = np.random.default_rng(seed)
    shapes = refnet.RefNet(arch, nfeat, nlabel)
    params = [torch.from_numpy(p).requires_grad_(True) for p in shapes.random_params(rng)]
    net = TorchNet(arch, nfeat, nlabel)
    x = torch.from_numpy(rng.normal(size=(B, 1, nfeat, T)).astype(np.float32))
    tgt = rng.integers(0, nlabel - 1, size=(B, L)).astype(np.int32)
    times = []
    for it in range(warmup + runs):
        for p in params:
            p.grad = None
        t0 = time.perf_counter()
        em = net.forward(x, params)
        emn = np.ascontiguousarray(em.detach().numpy())
        ctc = O.CTC(emn, tgt, scale_mode=4)
        ctc.forward()
        d_em = torch.from_numpy(np.ascontiguousarray(ctc.backward(), dtype=np.float32))
        em.backward(d_em)
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    return float(np.median(times)), times
