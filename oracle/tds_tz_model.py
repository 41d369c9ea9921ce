"""TEST INFRASTRUCTURE (oracle/): the INDEX ARITHMETIC of the block-Toeplitz TDS convolution kernels
(wav2letter_amd/csrc/conv_tds_tz.hpp, conv_tds_tzf.hpp) restated in float64 numpy, lane by lane: slab addresses, the
lane-half frame pairing and the split tail, the order of the B-operand registers, MFMA operand / accumulator ownership,
the store offsets and their range checks.  Never-written bytes are NaN, so a wrong address shows up as a NaN or a wrong
number.  Held against a direct convolution by tests/test_tds_tz_model.py (-m "not gpu"): the decomposition is checked on
the CPU before the HIP kernels are compared with the oracle on the device.

Reference operator: fl::Conv2D kw x 1 inside fl::TDSBlock (recipes/sota/2019/am_arch/am_tds_ctc.arch:7-37; data flow
recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70)."""
import numpy as np

KW = 21
HB = 16


def cfg(C, R, NCT):
    S = R + KW - 1
    C2 = C // 2
    SP = S // 2
    TAIL = S % 2
    TR = (C2 + 1) // 2 if TAIL else 0
    NRD = SP * C2 + TR
    RT = 4 // NCT
    GR = 2 * RT
    p = HB * C
    while (R * p) % 64 != 32:
        p += 4
    return dict(C=C, R=R, NCT=NCT, S=S, C2=C2, SP=SP, TR=TR, NRD=NRD, NK=2 * NRD, RT=RT, GR=GR, RF=GR * R, NF=(GR - 1) * R + S, PITCH=p)


CFGS = {10: cfg(10, 3, 1), 14: cfg(14, 2, 1), 18: cfg(18, 3, 2)}


def mfma_32x32x2(a, b, acc):
    """v_mfma_f32_32x32x2_f32: a[lane] = A[row lane & 31][k lane >> 5], b[lane] = B[k lane >> 5][col lane & 31],
    acc[v][lane] = D[row 8 (v >> 2) + 4 (lane >> 5) + (v & 3)][col lane & 31]"""
    A = np.stack([a[:32], a[32:]], axis=1)        # [32][2]
    B = np.stack([b[:32], b[32:]], axis=0)        # [2][32]
    D = A @ B
    lane = np.arange(64)
    for v in range(16):
        acc[v] += D[8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), lane & 31]
    return acc


def forward(x, w, bias, kw, padl, flip=False, relu=False, add=None, Tout=None):
    """x [B][Tin][H][C], w [kw][C][C]; returns y [B][Tout][H][C] computed the way tds_conv_tz_k does."""
    B, Tin, H, C = x.shape
    g = CFGS[C]
    R, NCT, S, C2, SP, TR, NRD, RT, RF, NF, PITCH = (g[k] for k in ("R", "NCT", "S", "C2", "SP", "TR", "NRD", "RT", "RF", "NF", "PITCH"))
    if Tout is None:
        Tout = Tin
    assert H % HB == 0 and kw <= KW
    y = np.full((B, Tout, H, C), np.nan)
    yflat = y.reshape(B, -1)
    xflat = x.reshape(B, -1).astype(np.float64)
    HC = H * C
    lane = np.arange(64)
    n, hf = lane & 31, lane >> 5
    rps = (Tout + RF - 1) // RF
    wf = w.astype(np.float64).reshape(-1)
    for b in range(B):
        for hb in range(H // HB):
            for k in range(rps):
                t0 = k * RF
                # ---- stage: NF frames, CPF 16-byte chunks each, buffer range check -> zeros; bytes behind the slab are zero
                slab = np.full(NF * PITCH + 16, np.nan)
                slab[NF * PITCH:] = 0.0
                for f in range(NF):
                    for piece in range(HB * C // 4):
                        off = ((t0 - padl + f) * HC + hb * HB * C) + piece * 4      # dwords
                        ok = 0 <= off and off + 4 <= Tin * HC
                        slab[f * PITCH + piece * 4: f * PITCH + piece * 4 + 4] = xflat[b, off:off + 4] if ok else 0.0
                for wave in range(4):
                    rt, ct = (wave, 0) if NCT == 1 else (wave >> 1, wave & 1)
                    nn = 32 * ct + n
                    rr, co = nn // C, nn % C
                    colOk = nn < R * C
                    # ---- B registers
                    bw = np.zeros((2 * NRD, 64))
                    wstep = 1 if flip else C
                    for sp in range(SP):
                        tap = 2 * sp + hf - rr
                        ok = colOk & (tap >= 0) & (tap < kw)
                        tc = np.where(ok, tap, 0)
                        base = ((kw - 1 - tc) * C + co) * C if flip else tc * C * C + co
                        for u in range(C):
                            bw[2 * sp * C2 + u] = np.where(ok, wf[base + u * wstep], 0.0)
                    if TR:
                        tap = S - 1 - rr
                        ok = colOk & (tap >= 0) & (tap < kw)
                        tc = np.where(ok, tap, 0)
                        base = ((kw - 1 - tc) * C + co) * C if flip else tc * C * C + co
                        for u in range(2 * TR):
                            ci = 2 * hf * TR + u
                            okc = ok & (ci < C)
                            bw[2 * SP * C2 + u] = np.where(okc, wf[base + np.where(okc, ci, 0) * wstep], 0.0)
                    rowOff = (R * (n >> 4) + 2 * R * rt) * PITCH + (n & 15) * C
                    aMain = rowOff + hf * PITCH
                    aTail = rowOff + hf * TR * 2
                    yLane = np.where(colOk, (rr + 2 * R * rt) * HC + 4 * hf * C + co, -(1 << 40))
                    yOff = yLane + t0 * HC + hb * HB * C
                    acc = np.zeros((16, 64))
                    voff = [yOff + (v >> 3) * R * HC + (8 * ((v >> 2) & 1) + (v & 3)) * C for v in range(16)]
                    if add is not None:
                        af = add.reshape(B, -1)
                        for v in range(16):
                            inr = (voff[v] >= 0) & (voff[v] < Tout * HC)
                            acc[v] = np.where(inr, af[b, np.clip(voff[v], 0, Tout * HC - 1)], 0.0)
                    for d in range(NRD):
                        if d < SP * C2:
                            sp, cp = divmod(d, C2)
                            addr = aMain + 2 * sp * PITCH + 2 * cp
                        else:
                            addr = aTail + (S - 1) * PITCH + 2 * (d - SP * C2)
                        acc = mfma_32x32x2(slab[addr], bw[2 * d], acc)
                        acc = mfma_32x32x2(slab[addr + 1], bw[2 * d + 1], acc)
                    if bias is not None:
                        acc = acc + np.where(colOk, bias.astype(np.float64)[co], 0.0)
                    if relu:
                        acc = np.maximum(acc, 0.0)
                    for v in range(16):
                        inr = (voff[v] >= 0) & (voff[v] < Tout * HC)
                        assert np.isnan(yflat[b, voff[v][inr]]).all(), "an output is written twice"
                        yflat[b, voff[v][inr]] = acc[v][inr]
    return y


def direct(x, w, bias, kw, padl, flip=False, relu=False, add=None, Tout=None):
    """out[t][h][co] = bias[co] + sum_{j, ci} x[t + j - padl][h][ci] W[j][ci][co]; flip: W'[j][ci][co] = W[kw-1-j][co][ci]"""
    B, Tin, H, C = x.shape
    if Tout is None:
        Tout = Tin
    ww = w.astype(np.float64)
    if flip:
        ww = ww[::-1].transpose(0, 2, 1)
    y = np.zeros((B, Tout, H, C))
    xp = np.zeros((B, Tout + kw - 1 + max(0, padl) + 64, H, C))
    for t in range(Tin):
        if 0 <= t + padl < xp.shape[1]:
            xp[:, t + padl] = x[:, t]
    for j in range(kw):
        y += np.einsum("bthc,cd->bthd", xp[:, j:j + Tout], ww[j])
    if bias is not None:
        y += bias
    if relu:
        y = np.maximum(y, 0)
    if add is not None:
        y += add
    return y


# ---------------------------------------------------------------------------------------------------------------------------
# filter gradient (conv_tds_tzf.hpp): D[(s, ci)][(r, co)] over (group i, mel row h), folded into dW by tds_tzf_reduce_k
def cfg_f(C, R, GR):
    S = R + KW - 1
    C2 = C // 2
    NPR = S * C2
    NPT = (NPR + 1 + 31) // 32
    p = HB * C + 4
    while p % 64 not in (40, 48, 8, 16, 24, 56):
        p += 4
    PD = HB * C + (12 if (HB * C) % 32 == 0 else 0)
    return dict(C=C, R=R, GR=GR, S=S, C2=C2, NPR=NPR, NPT=NPT, RF=GR * R, NFX=(GR - 1) * R + S, NFD=GR * R, PX=p, PD=PD)


CFGS_F = {10: cfg_f(10, 3, 16), 14: cfg_f(14, 2, 12)}


def filter_grad(x, dy, kw, padl, n_wg=3):
    """x [B][Tin][H][C], dy [B][Tout][H][C] -> (dW [kw][C][C], dbias [C]) computed the way tds_conv_tzf_k + tds_tzf_reduce_k do."""
    B, Tin, H, C = x.shape
    Tout = dy.shape[1]
    g = CFGS_F[C]
    R, GR, S, C2, NPR, NPT, RF, NFX, NFD, PX, PD = (g[k] for k in ("R", "GR", "S", "C2", "NPR", "NPT", "RF", "NFX", "NFD", "PX", "PD"))
    HC = H * C
    lane = np.arange(64)
    n, hf = lane & 31, lane >> 5
    xflat = x.reshape(B, -1).astype(np.float64)
    dflat = dy.reshape(B, -1).astype(np.float64)
    rps = (Tout + RF - 1) // RF
    rounds = [(b, hb, k) for b in range(B) for hb in range(H // HB) for k in range(rps)]
    rpw = (len(rounds) + n_wg - 1) // n_wg
    nblocks = (len(rounds) + rpw - 1) // rpw
    partial = np.zeros((nblocks, NPT, 2, 16, 64))
    for wg in range(nblocks):
        acc = np.zeros((8, NPT, 2, 16, 64))
        for (b, hb, k) in rounds[wg * rpw:(wg + 1) * rpw]:
            t0 = k * RF
            xs = np.full(NFX * PX, np.nan)
            ds = np.full(NFD * PD, np.nan)
            for f in range(NFX):
                xs[f * PX + HB * C: (f + 1) * PX] = 1.0
                for piece in range(HB * C // 4):
                    off = (t0 - padl + f) * HC + hb * HB * C + piece * 4
                    ok = 0 <= off and off + 4 <= Tin * HC
                    xs[f * PX + piece * 4: f * PX + piece * 4 + 4] = xflat[b, off:off + 4] if ok else 0.0
            for f in range(NFD):
                for piece in range(HB * C // 4):
                    off = (t0 + f) * HC + hb * HB * C + piece * 4
                    ok = 0 <= off and off + 4 <= Tout * HC
                    ds[f * PD + piece * 4: f * PD + piece * 4 + 4] = dflat[b, off:off + 4] if ok else 0.0
            for wave in range(8):
                rr = np.where(n < R * C, n // C, 0)
                co = np.where(n < R * C, n % C, 0)
                bBase = rr * PD + (2 * wave + hf) * C + co
                for i in range(GR):
                    bf = ds[bBase + i * R * PD]
                    for pt in range(NPT):
                        pr = 32 * pt + n
                        s, cp = pr // C2, pr % C2
                        aoff = np.where(pr < NPR, s * PX + (2 * wave + hf) * C + 2 * cp, HB * C) + i * R * PX
                        a0, a1 = xs[aoff], xs[aoff + 1]
                        a0 = np.where(pr > NPR, 0.0, a0)     # rows nobody exports: anything finite
                        a1 = np.where(pr > NPR, 0.0, a1)
                        acc[wave, pt, 0] = mfma_32x32x2(a0, bf, acc[wave, pt, 0])
                        acc[wave, pt, 1] = mfma_32x32x2(a1, bf, acc[wave, pt, 1])
        partial[wg] = acc.sum(axis=0)
    # ---- tds_tzf_reduce_k
    flat = partial.reshape(nblocks, -1)
    dW = np.zeros(kw * C * C)
    db = np.zeros(C)
    for o in range(kw * C * C + C):
        tot = 0.0
        for r in range(R):
            if o < kw * C * C:
                j, rem = divmod(o, C * C)
                ci, co = divmod(rem, C)
                pr, e = (j + r) * C2 + (ci >> 1), ci & 1
            else:
                co = o - kw * C * C
                pr, e = NPR, 0
            pt, row, col = pr >> 5, pr & 31, r * C + co
            idx = ((pt * 2 + e) * 16 + 4 * (row >> 3) + (row & 3)) * 64 + 32 * ((row >> 2) & 1) + col
            tot += flat[:, idx].sum()
        if o < kw * C * C:
            dW[o] = tot
        else:
            db[o - kw * C * C] = tot
    return dW.reshape(kw, C, C), db


def filter_direct(x, dy, kw, padl):
    B, Tin, H, C = x.shape
    Tout = dy.shape[1]
    dW = np.zeros((kw, C, C))
    for j in range(kw):
        for t in range(Tout):
            ti = t + j - padl
            if 0 <= ti < Tin:
                dW[j] += np.einsum("bhc,bhd->cd", x[:, ti].astype(np.float64), dy[:, t].astype(np.float64))
    return dW, dy.astype(np.float64).sum(axis=(0, 1, 2))
