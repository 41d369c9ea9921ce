"""Oracle-side composition of the reference modules (numpy glue over the C
oracle primitives), in the REFERENCE's layouts ([B][C][H][T] activations).
Test infrastructure only."""
import numpy as np

from oracle import pyoracle as O


def relu(x):
    return np.maximum(x, 0)


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


def tds_fwd(x, p, padl, padr, ln_mode="all", streaming=False, eps=1e-5, keep=False):
    B, Cc, H, T = x.shape
    a = O.conv_fwd(x, p.wc, p.bc, 1, padl, padr)
    r = relu(a) + x
    y = ln_fwd(r, ln_mode, float(p.g1), float(p.b1n), eps, streaming)
    z = to_frames(y)
    u = O.linear_fwd(z, p.w1, p.b1)
    v = O.linear_fwd(relu(u), p.w2, p.b2)
    s = from_frames(v, B, Cc, H, T) + y
    out = ln_fwd(s, ln_mode, float(p.g2), float(p.b2n), eps, streaming)
    if keep:
        return out, dict(x=x, a=a, r=r, y=y, z=z, u=u, s=s)
    return out


def tds_bwd(dout, p, saved, padl, padr, ln_mode="all", eps=1e-5):
    """returns dx and a dict of parameter grads"""
    x, a, r, y, z, u, s = (saved[k] for k in "x a r y z u s".split())
    B, Cc, H, T = x.shape
    g = {}
    ds, g["g2"], g["b2n"] = ln_bwd(s, dout, ln_mode, float(p.g2), eps)
    dy = ds.copy()
    dv = to_frames(ds)
    dru, g["w2"], g["b2"] = O.linear_bwd(relu(u), p.w2, dv)
    du = dru * (u > 0)
    dz, g["w1"], g["b1"] = O.linear_bwd(z, p.w1, du)
    dy += from_frames(dz, B, Cc, H, T)
    dr, g["g1"], g["b1n"] = ln_bwd(r, dy, ln_mode, float(p.g1), eps)
    da = dr * (a > 0)
    dxc, g["wc"], g["bc"] = O.conv_bwd(x, p.wc, da, 1, padl, padr)
    return dr + dxc, g
