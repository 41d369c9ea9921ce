"""TEST INFRASTRUCTURE (see oracle/criterion_oracle.c header): only tests/, smoke() and the
cpu_baseline leg of bench.py may import this.

Oracle-side composition of the reference modules (numpy glue over the C
oracle primitives), in the REFERENCE's layouts ([B][C][H][T] activations).
Test infrastructure only."""
import numpy as np

from oracle import pyoracle as O


def relu(x):
    return np.maximum(x, 0)


def bf16_round(a):
    """fp32 -> nearest bf16 (ties to even, what v_cvt_pk_bf16_f32 does) -> fp32: the value a bf16 GEMM operand carries"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32).reshape(np.shape(a))


# the (padded channels, k-steps, stride) triples the mixed-precision mode has convolution kernels for (conv_tds_bf16.hip):
# geometries outside stay in fp32 there, so the restatement leaves their operands unrounded too
_BF16_CONV_FWD = {(16, 9, 1), (16, 21, 1), (24, 15, 1), (24, 18, 1), (24, 33, 1), (32, 22, 1), (16, 5, 1), (16, 11, 1), (24, 9, 1),
                  (16, 10, 2), (24, 18, 2), (16, 21, 2)}
# filter gradient: (padded input channels, k-steps, stride, padded output channels)
_BF16_CONV_FILTER = {(16, 9, 1, 16), (16, 21, 1, 16), (24, 15, 1, 24), (24, 18, 1, 24), (24, 33, 1, 24), (32, 22, 1, 32), (16, 10, 2, 16),
                     (16, 10, 2, 24), (24, 18, 2, 24), (24, 18, 1, 32), (16, 21, 2, 16), (16, 21, 2, 24)}


def conv_rounds_to_bf16(cin, cout, kw, stride, H):
    """True when the product's mixed-precision mode multiplies this kw x 1 convolution (over H mel rows) in bf16"""
    if stride not in (1, 2) or not (1 <= cin <= 32 and 1 <= cout <= 32) or kw < stride or H % 16:
        return False
    cp = lambda c: 16 if c <= 16 else 24 if c <= 24 else 32

    def nstep(k, c):
        n = (k * c + 15) // 16
        while (n * 16) % c:
            n += 1
        return n
    f = (cp(cin), nstep(kw, cp(cin)), stride)
    b = (cp(cout), nstep((kw + stride - 1) // stride, cp(cout)), 1)
    return f in _BF16_CONV_FWD and b in _BF16_CONV_FWD and f + (cp(cout),) in _BF16_CONV_FILTER


def lin_fwd(x, w, b, bf16=False):
    """fl::Linear forward; bf16: both GEMM operands rounded to bf16 (fp32 accumulate, fp32 bias and result) -- the
    arithmetic of the mixed-precision mode (fl's AMP casts the operands of linear / conv, cpc/Train.cpp:1184 keeps the
    criterion input f32)"""
    if bf16:
        return O.linear_fwd(bf16_round(x), bf16_round(w), b)
    return O.linear_fwd(x, w, b)


def lin_bwd(x, w, dy, bf16=False):
    """(dx, dw, db); bf16: dx = bf(dy) bf(w)^T, dw = bf(x)^T bf(dy), db = column sums of bf(dy) (the bias gradient is a row of the
    weight-gradient product: a row of ones in the x^T image, host/net.cpp BfImage::onesRow)"""
    if bf16:
        return O.linear_bwd(bf16_round(x), bf16_round(w), bf16_round(dy))
    return O.linear_bwd(x, w, dy)


def ln_fwd(x, mode, gamma=1.0, beta=0.0, eps=1e-5, streaming=False):
    """x [B][C][H][T]; mode 'all' = LN axes {0,1,2} (per utterance), 'frame' = {1,2} (per frame)."""
    B, Cc, H, T = x.shape
    if mode == "all":
        return O.layernorm_fwd(x, B, gamma, beta, eps, streaming).reshape(x.shape)
    xf = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    y = O.layernorm_fwd(xf, B * T, gamma, beta, eps, streaming).reshape(B, T, Cc, H)
    return np.ascontiguousarray(y.transpose(0, 2, 3, 1))


def ln_bwd(x, dy, mode, gamma=1.0, eps=1e-5):
    B, Cc, H, T = x.shape
    if mode == "all":
        dx, dg, db = O.layernorm_bwd(x, dy, B, gamma, eps)
        return dx.reshape(x.shape), dg, db
    xf = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    df = np.ascontiguousarray(dy.transpose(0, 3, 1, 2))
    dx, dg, db = O.layernorm_bwd(xf, df, B * T, gamma, eps)
    return np.ascontiguousarray(dx.reshape(B, T, Cc, H).transpose(0, 2, 3, 1)), dg, db


def to_frames(x):
    """[B][C][H][T] -> [B*T][H*C] with feature f = h*C + c (Flashlight TDS: Reorder(2,1,0,3)+View)."""
    B, Cc, H, T = x.shape
    return np.ascontiguousarray(x.transpose(0, 3, 2, 1)).reshape(B * T, H * Cc)


def from_frames(z, B, Cc, H, T):
    return np.ascontiguousarray(z.reshape(B, T, H, Cc).transpose(0, 3, 2, 1))


class TDSParams:
    """10 parameters in the reference's order (StreamingTDSModelConverter.cpp:110-127):
    conv w [C][C][kw], conv b [C], ln1 gamma, beta, lin1 W [l][l2], b [l2], lin2 W [l2][l], b [l], ln2 gamma, beta"""

    def __init__(self, c, kw, h, l2=0, rng=None, scale=1.0):
        rng = rng or np.random.default_rng(0)
        l = c * h
        l2 = l2 or l
        self.c, self.kw, self.h, self.l, self.l2 = c, kw, h, l, l2
        u = lambda shape, fan: (rng.uniform(-1, 1, size=shape) * scale / np.sqrt(fan)).astype(np.float32)
        self.wc = u((c, c, kw), c * kw); self.bc = u((c,), c * kw)
        self.g1 = np.float32(1.0 + 0.1 * rng.normal()); self.b1n = np.float32(0.1 * rng.normal())
        self.w1 = u((l, l2), l); self.b1 = u((l2,), l)
        self.w2 = u((l2, l), l2); self.b2 = u((l,), l2)
        self.g2 = np.float32(1.0 + 0.1 * rng.normal()); self.b2n = np.float32(0.1 * rng.normal())


def tds_fwd(x, p, padl, padr, ln_mode="all", streaming=False, eps=1e-5, keep=False, bf16=False):
    """bf16: the operands of the block's convolution and of its two Linear layers are rounded to bf16 (fl's AMP casts the
    operands of conv2d and linear: recipes/slimIPL/src/Train.cpp:211, :1681-1760); accumulation, bias, ReLU, LayerNorm, fp32"""
    B, Cc, H, T = x.shape
    a = O.conv_fwd(bf16_round(x), bf16_round(p.wc), p.bc, 1, padl, padr) if bf16 else O.conv_fwd(x, p.wc, p.bc, 1, padl, padr)
    r = relu(a) + x
    y = ln_fwd(r, ln_mode, float(p.g1), float(p.b1n), eps, streaming)
    z = to_frames(y)
    u = lin_fwd(z, p.w1, p.b1, bf16)
    v = lin_fwd(relu(u), p.w2, p.b2, bf16)
    s = from_frames(v, B, Cc, H, T) + y
    out = ln_fwd(s, ln_mode, float(p.g2), float(p.b2n), eps, streaming)
    if keep:
        return out, dict(x=x, a=a, r=r, y=y, z=z, u=u, s=s)
    return out


def tds_bwd(dout, p, saved, padl, padr, ln_mode="all", eps=1e-5, bf16=False):
    """returns dx and a dict of parameter grads"""
    x, a, r, y, z, u, s = (saved[k] for k in "x a r y z u s".split())
    B, Cc, H, T = x.shape
    g = {}
    ds, g["g2"], g["b2n"] = ln_bwd(s, dout, ln_mode, float(p.g2), eps)
    dy = ds.copy()
    dv = to_frames(ds)
    dru, g["w2"], g["b2"] = lin_bwd(relu(u), p.w2, dv, bf16)
    du = dru * (u > 0)
    dz, g["w1"], g["b1"] = lin_bwd(z, p.w1, du, bf16)
    dy += from_frames(dz, B, Cc, H, T)
    dr, g["g1"], g["b1n"] = ln_bwd(r, dy, ln_mode, float(p.g1), eps)
    da = dr * (a > 0)
    if bf16:   # dx = conv^T(bf(da), bf(w)), dw = bf(x) (*) bf(da), db = sums of bf(da) (summed from the slabs of the filter-gradient kernel)
        dxc, g["wc"], dbr = O.conv_bwd(bf16_round(x), bf16_round(p.wc), bf16_round(da), 1, padl, padr)
        g["bc"] = dbr
    else:
        dxc, g["wc"], g["bc"] = O.conv_bwd(x, p.wc, da, 1, padl, padr)
    return dr + dxc, g


# ---------------------------------------------------------------------------------------------
# Reference-layout interpreter of an arch text (ArrayFire dims (d0,d1,d2,d3) stored as a numpy
# array of shape (d3,d2,d1,d0)), forward and backward through the oracle primitives.  Parameters
# are a list in Flashlight params() order, in Flashlight memory layouts:
#   conv w [cout][cin][kw], conv b [cout], linear W [in][out], b [out], LN (gamma, beta) pair,
#   WeightNorm: v (as the wrapped weight), g [nout], (bias)
class RefNet:
    def __init__(self, arch_text, nfeat, nlabel, bf16=False):
        """bf16: every fl::Linear (stand-alone `L` lines and the two inside a TDS block) multiplies bf16-rounded operands
        -- the reference side of the mixed-precision parity tests (BASELINE config 3)"""
        self.bf16 = bf16
        self.lines = []
        for raw in arch_text.splitlines():
            l = raw.strip()
            if not l or l.startswith("#"):
                continue
            self.lines.append(l.replace("NFEAT", str(nfeat)).replace("NLABEL", str(nlabel)).split())
        self.nfeat = nfeat

    def param_shapes(self):
        """list of (kind, shape) in params() order"""
        out = []
        for f in self.lines:
            t = f
            wn = False
            if t[0] == "WN":
                wn, t = True, t[2:]
            if t[0] in ("C", "C1"):
                cin, cout, kw = int(t[1]), int(t[2]), int(t[3])
                out.append(("conv.w", (cout, cin, kw)))
                if wn:
                    out.append(("wn.g", (cout,)))
                out.append(("conv.b", (cout,)))
            elif t[0] == "C2":
                cin, cout, kw, kh = int(t[1]), int(t[2]), int(t[3]), int(t[4])
                # Flashlight Conv2D weight, ArrayFire dims (kw, kh, cin, cout) == row-major [cout][cin][kh][kw]
                out += [("conv.w", (cout, cin, kw) if kh == 1 else (cout, cin, kh, kw)), ("conv.b", (cout,))]
            elif t[0] == "L":
                out.append(("linear.w", (int(t[1]), int(t[2]))))
                if wn:
                    out.append(("wn.g", (int(t[2]),)))
                out.append(("linear.b", (int(t[2]),)))
            elif t[0] == "LN":
                out.append(("ln", (2,)))
            elif t[0] == "TDS":
                c, kw, h = int(t[1]), int(t[2]), int(t[3])
                l = c * h
                l2 = int(t[5]) if len(t) > 5 and int(t[5]) else l
                out += [("conv.w", (c, c, kw)), ("conv.b", (c,)), ("ln", (2,)), ("linear.w", (l, l2)), ("linear.b", (l2,)),
                        ("linear.w", (l2, l)), ("linear.b", (l,)), ("ln", (2,))]
            elif t[0] == "TR":
                from oracle import transformer_oracle as TO
                out += TO.tr_param_shapes(int(t[1]), int(t[2]), int(t[3]), int(t[4]))
        return out

    def random_params(self, rng):
        ps = []
        for kind, shape in self.param_shapes():
            if kind == "ln":
                ps.append(np.array([1 + 0.1 * rng.normal(), 0.1 * rng.normal()], np.float32))
            elif kind == "wn.g":
                ps.append((1 + 0.2 * rng.normal(size=shape)).astype(np.float32))
            elif kind.endswith(".w"):
                fan = shape[1] * shape[2] if kind == "conv.w" else shape[0]
                ps.append((rng.uniform(-1, 1, size=shape) * np.sqrt(3.0 / fan)).astype(np.float32))
            else:
                ps.append((rng.uniform(-1, 1, size=shape) * 0.3).astype(np.float32))
        return ps

    @staticmethod
    def _conv_pads(t, T):
        """(cin, cout, kw, stride, padl, padr) of a C / C2 line"""
        if t[0] == "C2":
            cin, cout, kw, stride = int(t[1]), int(t[2]), int(t[3]), int(t[5])
            pad = int(t[7]) if len(t) > 7 else 0
        else:
            cin, cout, kw, stride = int(t[1]), int(t[2]), int(t[3]), int(t[4])
            pad = int(t[5]) if len(t) > 5 else 0
        if pad == -1:
            pad = O.same_pad(T, kw, stride)
        return cin, cout, kw, stride, pad, pad

    def forward(self, x, params):
        """x: numpy [B][1][NFEAT][T] (af dims (T, NFEAT, 1, B)). returns emissions [B][T'][N]"""
        a = x
        pi = 0
        self.tape = []
        self.input_frames = x.shape[3]
        for f in self.lines:
            t = f
            wn_dim = None
            if t[0] == "WN":
                wn_dim, t = int(t[1]), t[2:]
            if t[0] == "SAUG":
                continue
            if t[0] == "PD":
                # fl::Padding(val, {l0, r0}, ...) on ArrayFire dim 0 = time (streaming arch: asymmetric padding ahead of an
                # unpadded convolution, am_500ms_future_context.arch:3)
                assert float(t[1]) == 0.0 and all(int(v) == 0 for v in t[4:]), t
                l0, r0 = int(t[2]), int(t[3])
                self.tape.append(("PD", l0, a.shape[3]))
                a = np.ascontiguousarray(np.pad(a, ((0, 0), (0, 0), (0, 0), (l0, r0))))
                continue
            if t[0] == "V":
                dims = [int(v) for v in t[1:5]]
                cur = list(a.shape[::-1])
                for i in range(4):
                    if dims[i] == 0:
                        dims[i] = cur[i]
                if -1 in dims:
                    k = dims.index(-1)
                    dims[k] = int(a.size // np.prod([d for d in dims if d != -1]))
                self.tape.append(("V", a.shape))
                a = np.ascontiguousarray(a).reshape(dims[::-1])
            elif t[0] == "RO":
                p = [int(v) for v in t[1:5]]
                axes = [0] * 4
                for i in range(4):
                    axes[3 - i] = 3 - p[i]
                self.tape.append(("RO", axes))
                a = np.ascontiguousarray(a.transpose(axes))
            elif t[0] in ("C", "C1", "C2"):
                T = a.shape[3]
                cin, cout, kw, stride, pl, pr = self._conv_pads(t, T)
                w = params[pi]; pi += 1
                kh = int(t[4]) if t[0] == "C2" else 1
                if kh > 1:
                    # kh x kw kernel, SAME on the mel axis (am_tds_ctc_librivox.arch): a kw x 1 convolution over kh*cin
                    # channels, channel dh*cin + ci of mel row h = input row h + dh - (kh-1)/2 (zero outside)
                    assert wn_dim is None and kh % 2 == 1 and (len(t) <= 8 or int(t[8]) in (-1, (kh - 1) // 2)), t
                    B_, _, H_, T_ = a.shape
                    ph = (kh - 1) // 2
                    ap = np.zeros((B_, cin, H_ + 2 * ph, T_), a.dtype)
                    ap[:, :, ph:ph + H_] = a
                    ae = np.concatenate([ap[:, :, dh:dh + H_] for dh in range(kh)], axis=1)
                    we = np.ascontiguousarray(w.transpose(0, 2, 1, 3)).reshape(cout, kh * cin, kw)
                    b = params[pi]; pi += 1
                    self.tape.append(("C2D", ae, we, stride, pl, pr, pi, kh, cin))
                    a = O.conv_fwd(np.ascontiguousarray(ae), we, b, stride, pl, pr)
                    continue
                v = w
                g = None
                if wn_dim is not None:
                    g = params[pi]; pi += 1
                    w = O.weightnorm_fwd(v, g, 1, cout, cin * kw).reshape(v.shape)
                b = params[pi]; pi += 1
                rnd = self.bf16 and conv_rounds_to_bf16(cin, cout, kw, stride, a.shape[2])
                if rnd:   # the mixed-precision mode's sub-sampling convolutions: bf16 operands, fp32 accumulation and bias
                    self.tape.append(("C", bf16_round(a), bf16_round(w), v, g, stride, pl, pr, pi, True))
                    a = O.conv_fwd(bf16_round(a), bf16_round(w), b, stride, pl, pr)
                    continue
                self.tape.append(("C", a, w, v, g, stride, pl, pr, pi, False))
                a = O.conv_fwd(a, w, b, stride, pl, pr)
            elif t[0] == "L":
                nin, nout = int(t[1]), int(t[2])
                w = params[pi]; pi += 1
                v, g = w, None
                if wn_dim is not None:
                    g = params[pi]; pi += 1
                    w = O.weightnorm_fwd(v, g, nin, nout, 1).reshape(v.shape)
                b = params[pi]; pi += 1
                shp = a.shape
                assert shp[3] == nin, (shp, nin)
                z = np.ascontiguousarray(a).reshape(-1, nin)
                self.tape.append(("L", z, w, v, g, shp, pi))
                a = lin_fwd(z, w, b, self.bf16).reshape(shp[:3] + (nout,))
            elif t[0] == "R":
                self.tape.append(("R", a))
                a = relu(a)
            elif t[0] == "DO":
                assert float(t[1]) == 0.0, "reference interpreter runs dropout-free archs"
            elif t[0] == "LN":
                axes = sorted(int(v) for v in t[1:])
                mode = "all" if axes == [0, 1, 2] else "frame"
                gb = params[pi]; pi += 1
                self.tape.append(("LN", a, mode, gb, pi))
                a = ln_fwd(a, mode, float(gb[0]), float(gb[1]))
            elif t[0] == "GLU":
                d = int(t[1])
                ax = 3 - d
                outer = int(np.prod(a.shape[:ax])); half = a.shape[ax] // 2; inner = int(np.prod(a.shape[ax + 1:]))
                self.tape.append(("GLU", a, outer, half, inner))
                shp = list(a.shape); shp[ax] = half
                a = O.glu_fwd(a, outer, half, inner).reshape(shp)
            elif t[0] == "TDS":
                c, kw, h = int(t[1]), int(t[2]), int(t[3])
                assert (len(t) <= 4 or float(t[4]) == 0.0)
                l = c * h
                l2 = int(t[5]) if len(t) > 5 and int(t[5]) else l
                p = TDSParams(c, kw, h, l2)
                p.wc, p.bc = params[pi], params[pi + 1]
                p.g1, p.b1n = params[pi + 2]
                p.w1, p.b1, p.w2, p.b2 = params[pi + 3:pi + 7]
                p.g2, p.b2n = params[pi + 7]
                pi += 8
                rpad = int(t[6]) if len(t) > 6 else -1
                if rpad < 0:
                    pl = pr = O.same_pad(a.shape[3], kw, 1)
                else:
                    pr, pl = rpad, kw - 1 - rpad
                mode = "frame" if (len(t) > 7 and int(t[7]) == 0) else "all"
                out, saved = tds_fwd(a, p, pl, pr, mode, keep=True, bf16=self.bf16)
                self.tape.append(("TDS", p, saved, pl, pr, mode, pi))
                a = out
            elif t[0] == "M":
                # fl::Pool2D(wx, 1, sx, 1, MAX) over time (SequentialBuilder.cpp:398-414)
                wx, wy, sx, sy = (int(v) for v in t[1:5])
                assert wy == 1 and sy == 1 and all(int(v) == 0 for v in t[5:]), t
                To = (a.shape[3] - wx) // sx + 1
                win = np.stack([a[..., k:k + (To - 1) * sx + 1:sx] for k in range(wx)], axis=0)
                arg = win.argmax(axis=0)          # first maximum, as a scan over the window finds it
                self.tape.append(("M", a.shape, arg, sx))
                a = np.ascontiguousarray(win.max(axis=0))
            elif t[0] == "TR":
                import torch
                from oracle import transformer_oracle as TO
                C, mlp, nheads, csz = int(t[1]), int(t[2]), int(t[3]), int(t[4])
                # training-mode randomness is supplied by the caller: self.tr_opts = [{"attn_mask": ..., "f": ...}, ...] per block
                opts = getattr(self, "tr_opts", None)
                opt = opts[sum(1 for r in self.tape if r[0] == "TR")] if opts else {}
                assert opts or (float(t[5]) == 0.0 and (len(t) <= 6 or float(t[6]) == 0.0)), "dropout needs the caller's masks"
                n = len(TO.tr_param_shapes(C, mlp, nheads, csz))
                assert a.shape[0] == 1 and a.shape[3] == C, a.shape     # (C, T, B, 1)
                xt = torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
                pt = [torch.tensor(np.asarray(p), dtype=torch.float64, requires_grad=True) for p in params[pi:pi + n]]
                pi += n
                am = opt.get("attn_mask")
                kl = None
                if getattr(self, "input_sizes", None) is not None:   # set by the caller: per-utterance input sizes of the batch
                    kl = TO.key_lengths(self.input_sizes, self.input_frames, xt.shape[1])
                yt = TO.tr_block(xt, pt, nheads, csz, None if am is None else torch.tensor(am, dtype=torch.float64), opt.get("f", 1.0), kl)
                self.tape.append(("TR", xt, pt, yt, pi))
                a = yt.detach().numpy().astype(np.float32)[None]
            else:
                raise ValueError(t[0])
        # (N, T, B, 1) -> [B][T][N]
        assert a.shape[0] == 1, a.shape
        return np.ascontiguousarray(a[0])

    def backward(self, d_em, nparams):
        """d_em [B][T'][N]; returns list of parameter gradients (Flashlight layouts)"""
        g = [None] * nparams
        da = d_em[None]
        for rec in reversed(self.tape):
            k = rec[0]
            if k in ("TDS", "TR") and getattr(self, "upstream", None) is not None:
                # teacher-forced per-block tests: the gradient arriving at this block's output, beside the record that holds its input
                self.upstream.append((rec, np.array(da, dtype=np.float32, copy=True)))
            if k == "PD":
                da = np.ascontiguousarray(da[..., rec[1]:rec[1] + rec[2]])
            elif k == "V":
                da = np.ascontiguousarray(da).reshape(rec[1])
            elif k == "RO":
                inv = np.argsort(rec[1])
                da = np.ascontiguousarray(da.transpose(inv))
            elif k == "C":
                _, a, w, v, gg, stride, pl, pr, pi, rnd = rec
                if rnd:   # dx, dw and the bias gradient from the rounded gradient
                    da = np.ascontiguousarray(da, dtype=np.float32)
                    dx, dw, db = O.conv_bwd(a, w, bf16_round(da), stride, pl, pr)
                else:
                    dx, dw, db = O.conv_bwd(a, w, da, stride, pl, pr)
                g[pi - 1] = db
                if gg is not None:
                    cout = w.shape[0]
                    dv, dg = O.weightnorm_bwd(v, gg, dw, 1, cout, v.size // cout)
                    g[pi - 2] = dg; g[pi - 3] = dv.reshape(v.shape)
                else:
                    g[pi - 2] = dw
                da = dx
            elif k == "C2D":
                _, ae, we, stride, pl, pr, pi, kh, cin = rec
                dxe, dwe, db = O.conv_bwd(np.ascontiguousarray(ae), we, da, stride, pl, pr)
                g[pi - 1] = db
                cout = we.shape[0]
                g[pi - 2] = np.ascontiguousarray(dwe.reshape(cout, kh, cin, -1).transpose(0, 2, 1, 3))
                B_, _, H_, T_ = dxe.shape
                ph = (kh - 1) // 2
                dp = np.zeros((B_, cin, H_ + 2 * ph, T_), dxe.dtype)
                for dh in range(kh):
                    dp[:, :, dh:dh + H_] += dxe[:, dh * cin:(dh + 1) * cin]
                da = dp[:, :, ph:ph + H_]
            elif k == "L":
                _, z, w, v, gg, shp, pi = rec
                dz, dw, db = lin_bwd(z, w, np.ascontiguousarray(da, dtype=np.float32).reshape(z.shape[0], -1), self.bf16)
                g[pi - 1] = db
                if gg is not None:
                    dv, dg = O.weightnorm_bwd(v, gg, dw, v.shape[0], v.shape[1], 1)
                    g[pi - 2] = dg; g[pi - 3] = dv.reshape(v.shape)
                else:
                    g[pi - 2] = dw
                da = dz.reshape(shp)
            elif k == "R":
                da = da * (rec[1] > 0)
            elif k == "LN":
                _, a, mode, gb, pi = rec
                da, dg, db = ln_bwd(a, np.ascontiguousarray(da, dtype=np.float32), mode, float(gb[0]))
                g[pi - 1] = np.array([dg, db], np.float32)
            elif k == "GLU":
                _, a, outer, half, inner = rec
                da = O.glu_bwd(a, np.ascontiguousarray(da, dtype=np.float32), outer, half, inner).reshape(a.shape)
            elif k == "M":
                _, shp, arg, sx = rec
                dx = np.zeros(shp, np.float32)
                To = arg.shape[3]
                tpos = arg + (np.arange(To) * sx)[None, None, None, :]
                b, c, h, _ = np.indices(arg.shape)
                np.add.at(dx, (b, c, h, tpos), np.asarray(da, np.float32))
                da = dx
            elif k == "TR":
                import torch
                _, xt, pt, yt, pi = rec
                gs = torch.autograd.grad(yt, [xt] + pt, grad_outputs=torch.tensor(np.asarray(da[0]), dtype=torch.float64),
                                         allow_unused=True)
                gs = [torch.zeros_like(v) if gq is None else gq for gq, v in zip(gs, [xt] + pt)]
                da = gs[0].numpy().astype(np.float32)[None]
                for j, gj in enumerate(gs[1:]):
                    g[pi - len(pt) + j] = gj.numpy().astype(np.float32)
            elif k == "TDS":
                _, p, saved, pl, pr, mode, pi = rec
                da, gg = tds_bwd(np.ascontiguousarray(da, dtype=np.float32), p, saved, pl, pr, mode, bf16=self.bf16)
                base = pi - 8
                g[base], g[base + 1] = gg["wc"], gg["bc"]
                g[base + 2] = np.array([gg["g1"], gg["b1n"]], np.float32)
                g[base + 3], g[base + 4], g[base + 5], g[base + 6] = gg["w1"], gg["b1"], gg["w2"], gg["b2"]
                g[base + 7] = np.array([gg["g2"], gg["b2n"]], np.float32)
        return g
