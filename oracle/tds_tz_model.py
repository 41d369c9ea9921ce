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


def cfg(CI, CO, R, NCT, SIG=1, KWM=21, ST=1):
    S = SIG * (R - 1) + KWM
    C2 = CI // 2
    SSPLIT = 1 if (NCT == 2 and R == 3) else 0
    SW = S - SIG * SSPLIT
    SP = SW // 2
    TAIL = SW % 2
    TR = (C2 + 1) // 2 if TAIL else 0
    NRD = SP * C2 + TR
    RT = 4 // NCT
    GR = 2 * RT
    GSTEP = SIG * R
    NF = (GSTEP * (GR - 1) + S + 3) // 4 * 4
    p = HB * CI
    while (GSTEP * p) % 64 != 32:
        p += 4
    if 4 * ((NF * p * 4 + 64 + 1023) // 1024 * 1024) > 160 * 1024:
        p = HB * CI
    # tap pitch of the weight copy in LDS: a lane's gather address is (its column's frame shift) * pitch + co, and the pitch is
    # chosen so that the shift of output frame rr moves the bank by rr CO: the 32 lanes of a half then sit on banks nn mod 32
    def wp(mult, target):
        q = CI * CO
        while (mult * q) % 32 != target:
            q += 1
        return q
    return dict(CI=CI, CO=CO, R=R, NCT=NCT, SIG=SIG, KWM=KWM, ST=ST, S=S, SW=SW, SSPLIT=SSPLIT, C2=C2, SP=SP, TR=TR, NRD=NRD, NK=2 * NRD, RT=RT, GR=GR, RF=GR * R,
                GSTEP=GSTEP, NF=NF, PITCH=p, NQF=S + SIG * (R - 1), NQB=ST * (S + R - 1), WPF=wp(SIG, (-CO) % 32), WPB=wp(ST, CO % 32))


# the instances conv_tds_rs.hip launches: (CI, CO, stride of the layer / tap step of the phase, backward?)
CFGS = {(10, 10, 1, False): cfg(10, 10, 3, 1), (14, 14, 1, False): cfg(14, 14, 2, 1), (18, 18, 1, False): cfg(18, 18, 3, 2),
        (10, 10, 1, True): cfg(10, 10, 3, 1), (14, 14, 1, True): cfg(14, 14, 2, 1), (18, 18, 1, True): cfg(18, 18, 3, 2),
        (10, 14, 2, False): cfg(10, 14, 2, 1, SIG=2), (14, 18, 2, False): cfg(14, 18, 3, 2, SIG=2),
        (14, 10, 2, True): cfg(14, 10, 3, 1, KWM=11, ST=2), (18, 14, 2, True): cfg(18, 14, 2, 1, KWM=11, ST=2)}


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


def launch(x, w, bias, kw, padl, y, g, flip=False, relu=False, add=None, Tout=None, tapOff=0, oOff=0, oStep=1):
    """one launch of tds_conv_tz_k<g>: x [B][Tin][H][CI], w the layer's weights [kwFull][.][.] (forward: [tap][ci][co]; backward:
    read as w[tapOff + ST (kw - 1 - tap)][co][ci]); writes frames oOff + oStep u (u < Tout) of y [B][ToutFull][H][CO] (NaN =
    never written)."""
    B, Tin, H, CI = x.shape
    CO, R, NCT, SIG, ST, S, SW, SSPLIT, C2, SP, TR, NRD, RT, RF, GSTEP, NF, PITCH = (g[k] for k in (
        "CO", "R", "NCT", "SIG", "ST", "S", "SW", "SSPLIT", "C2", "SP", "TR", "NRD", "RT", "RF", "GSTEP", "NF", "PITCH"))
    assert CI == g["CI"] and H % HB == 0 and kw <= g["KWM"] and y.shape[3] == CO
    ToutFull = y.shape[1]
    yflat = y.reshape(B, -1)
    xflat = x.reshape(B, -1).astype(np.float64)
    HCI, HCO = H * CI, H * CO
    CC = CI * CO
    lane = np.arange(64)
    n, hf = lane & 31, lane >> 5
    rps = (Tout + RF - 1) // RF
    wf = w.astype(np.float64).reshape(-1)
    kwFull = w.shape[0]
    # ---- the zero-padded weight copy of the prologue (LDS, second slab): [tap][ci][co] at the padded tap pitch WP, whatever
    # the order in HBM (the backward pass reads w[tap][co][ci]: transposed on the way through the registers); bytes the copy
    # does not write are NaN
    NQ = g["NQB"] if flip else g["NQF"]
    P = ST * (S - kw) if flip else SIG * (R - 1)
    WP = g["WPB"] if flip else g["WPF"]
    assert NQ * WP * 4 <= (NF * PITCH * 4 + 64 + 1023) // 1024 * 1024, "the weight copy fits the second slab"
    wl = np.full(NQ * WP, np.nan)
    for e in range(NQ * CC):
        src = e - P * CC
        v = wf[src] if 0 <= src < kwFull * CC else 0.0
        tap, r = divmod(e, CC)
        if flip:
            co_, ci_ = divmod(r, CI)
            wl[tap * WP + ci_ * CO + co_] = v
        else:
            wl[tap * WP + r] = v

    def gather(addr, colOk):
        """one ds_read_b32 of the gather: conflict-free = the valid columns of a lane half on pairwise different banks"""
        for h in (0, 1):
            a = addr[(hf == h) & colOk]
            assert len(set(a % 32)) == len(set(a)), "bank conflict in the weight gather"
        return wl[addr]
    for b in range(B):
        for hb in range(H // HB):
            for k in range(rps):
                # ---- stage: NF frames, buffer range check -> zeros; the bytes behind the slab are zero
                slab = np.full(NF * PITCH + 16, np.nan)
                slab[NF * PITCH:] = 0.0
                for f in range(NF):
                    if TR and PITCH > HB * CI:
                        slab[f * PITCH + HB * CI: (f + 1) * PITCH] = 0.0     # zeroed once by the kernel's prologue
                    for piece in range(HB * CI // 4):
                        off = ((k * RF * SIG - padl + f) * HCI + hb * HB * CI) + piece * 4      # dwords
                        ok = 0 <= off and off + 4 <= Tin * HCI
                        slab[f * PITCH + piece * 4: f * PITCH + piece * 4 + 4] = xflat[b, off:off + 4] if ok else 0.0
                for wave in range(4):
                    rt, ct = (wave, 0) if NCT == 1 else (wave >> 1, wave & 1)
                    nn = 32 * ct + n
                    colOk = nn < R * CO
                    s0 = SIG * ct if SSPLIT else 0
                    rr = np.where(colOk, nn // CO, ct if SSPLIT else 0)
                    co = np.where(colOk, nn % CO, 0)
                    # ---- B registers: gathers out of the LDS copy at one base + immediates
                    bw = np.zeros((2 * NRD, 64))
                    if not flip:
                        bm = (s0 + hf - SIG * rr + P) * WP + co
                        assert (bm >= 0).all()
                        for sp in range(SP):
                            for u in range(CI):
                                bw[2 * sp * C2 + u] = gather(bm + 2 * sp * WP + u * CO, colOk)
                        if TR:
                            bt = (s0 + SW - 1 - SIG * rr + P) * WP + 2 * hf * TR * CO + co
                            for u in range(2 * TR):
                                pad = 2 * TR + u >= CI          # ci = 2 hf TR + u >= CI in half 1: the padding pair
                                t = gather(np.where(pad & (hf == 1), bt - 2 * TR * CO, bt) + u * CO, colOk)
                                if pad:
                                    t = np.where(hf == 1, 0.0, t)
                                bw[2 * SP * C2 + u] = t
                    else:
                        bm = (tapOff + ST * (kw - 1 - hf - s0 + rr) + P - 2 * ST * (SP - 1)) * WP + co
                        assert (bm >= 0).all()
                        for sp in range(SP):
                            for u in range(CI):
                                bw[2 * sp * C2 + u] = gather(bm + 2 * ST * (SP - 1 - sp) * WP + u * CO, colOk)
                        if TR:
                            bt = (tapOff + ST * (kw - s0 - SW + rr) + P) * WP + co + 2 * hf * TR * CO
                            assert (bt >= 0).all()
                            for u in range(2 * TR):
                                pad = 2 * TR + u >= CI
                                t = gather(np.where(pad & (hf == 1), bt - 2 * TR * CO, bt) + u * CO, colOk)
                                if pad:
                                    t = np.where(hf == 1, 0.0, t)
                                bw[2 * SP * C2 + u] = t
                    rowOff = (GSTEP * (n >> 4) + 2 * GSTEP * rt) * PITCH + (n & 15) * CI
                    aMain = rowOff + (s0 + hf) * PITCH
                    aTail = rowOff + s0 * PITCH + hf * TR * 2
                    yLane = np.where(colOk, (rr + 2 * R * rt) * oStep * HCO + 4 * hf * CO + co, -(1 << 40))
                    g0 = yLane + (k * RF * oStep + oOff) * HCO + hb * HB * CO
                    acc = np.zeros((16, 64))
                    voff = [g0 + (v >> 3) * R * oStep * HCO + (8 * ((v >> 2) & 1) + (v & 3)) * CO for v in range(16)]
                    if add is not None:
                        af = add.reshape(B, -1)
                        for v in range(16):
                            inr = (voff[v] >= 0) & (voff[v] < ToutFull * HCO)
                            acc[v] = np.where(inr, af[b, np.clip(voff[v], 0, ToutFull * HCO - 1)], 0.0)
                    for d in range(NRD):
                        if d < SP * C2:
                            sp, cp = divmod(d, C2)
                            addr = aMain + 2 * sp * PITCH + 2 * cp
                        else:
                            addr = aTail + (SW - 1) * PITCH + 2 * (d - SP * C2)
                        acc = mfma_32x32x2(slab[addr], bw[2 * d], acc)
                        acc = mfma_32x32x2(slab[addr + 1], bw[2 * d + 1], acc)
                    acc = np.where(colOk, acc, 0.0)            # (the padding columns: something nobody stores)
                    if bias is not None:
                        acc = acc + np.where(colOk, bias.astype(np.float64)[co], 0.0)
                    if relu:
                        acc = np.maximum(acc, 0.0)
                    for v in range(16):
                        inr = (voff[v] >= 0) & (voff[v] < ToutFull * HCO)
                        assert np.isnan(yflat[b, voff[v][inr]]).all(), "an output is written twice"
                        yflat[b, voff[v][inr]] = acc[v][inr]
    return y


def forward(x, w, bias, kw, padl, flip=False, relu=False, add=None, Tout=None, stride=1):
    """the layer's forward pass (stride 1 or 2), or with flip its stride-1 backward-data pass, the way tds_conv_tz_k does it"""
    B, Tin, H, CI = x.shape
    CO = w.shape[1] if flip else w.shape[2]
    if Tout is None:
        Tout = Tin
    y = np.full((B, Tout, H, CO), np.nan)
    return launch(x, w, bias, kw, padl, y, CFGS[(CI, CO, stride, flip)], flip, relu, add, Tout)


def backward_data_strided(dy, w, T, kw, stride, padl, add=None):
    """dx [B][T][H][Cin] of a stride-2 layer from dy [B][To][H][Cout]: one launch per phase of the stride, as
    tds_conv_backward_data (conv_tds.hip) cuts it"""
    B, To, H, Cout = dy.shape
    Cin = w.shape[1]
    dx = np.full((B, T, H, Cin), np.nan)
    for f in range(stride):
        kwf = (kw - f + stride - 1) // stride
        c0 = (f - padl) % stride
        if c0 >= T:
            continue
        U = (T - c0 + stride - 1) // stride
        s0 = (c0 + padl - f) // stride
        launch(dy, w, None, kwf, kwf - 1 - s0, dx, CFGS[(Cout, Cin, stride, True)], True, False, add, U, tapOff=f, oOff=c0, oStep=stride)
    return dx


def direct(x, w, bias, kw, padl, flip=False, relu=False, add=None, Tout=None, stride=1):
    """out[t][h][co] = bias[co] + sum_{j, ci} x[stride t + j - padl][h][ci] W[j][ci][co]; flip: W'[j][ci][co] = W[kw-1-j][co][ci]"""
    B, Tin, H, CI = x.shape
    if Tout is None:
        Tout = Tin
    ww = w.astype(np.float64)
    if flip:
        ww = ww[::-1].transpose(0, 2, 1)
    CO = ww.shape[2]
    y = np.zeros((B, Tout, H, CO))
    for t in range(Tout):
        for j in range(kw):
            ti = stride * t + j - padl
            if 0 <= ti < Tin:
                y[:, t] += np.einsum("bhc,cd->bhd", x[:, ti].astype(np.float64), ww[j])
    if bias is not None:
        y += bias
    if relu:
        y = np.maximum(y, 0)
    if add is not None:
        y += add
    return y


def direct_backward_data(dy, w, T, kw, stride, padl, add=None):
    """dx[ti][h][ci] = sum_{to, j: stride to + j - padl = ti} dy[to][h][co] W[j][ci][co]"""
    B, To, H, Cout = dy.shape
    Cin = w.shape[1]
    dx = np.zeros((B, T, H, Cin))
    for to in range(To):
        for j in range(kw):
            ti = stride * to + j - padl
            if 0 <= ti < T:
                dx[:, ti] += np.einsum("bhd,cd->bhc", dy[:, to].astype(np.float64), w[j].astype(np.float64))
    if add is not None:
        dx += add
    return dx


# ---------------------------------------------------------------------------------------------------------------------------
# filter gradient (conv_tds_tzf.hpp): D[(s, ci)][(r, co)] over (group i, mel row h), folded into dW by tds_tzf_reduce_k
def cfg_f(CI, CO, R, GR, SIG=1):
    S = SIG * (R - 1) + KW
    C2 = CI // 2
    NPR = S * C2
    NPT = (NPR + 1 + 31) // 32
    p = HB * CI + 4
    while p % 64 not in (40, 48, 8, 16, 24, 56):
        p += 4
    PD = HB * CO + (12 if (HB * CO) % 32 == 0 else 0)
    return dict(CI=CI, CO=CO, R=R, GR=GR, SIG=SIG, S=S, C2=C2, NPR=NPR, NPT=NPT, RF=GR * R, NFX=SIG * R * (GR - 1) + S, NFD=GR * R, PX=p, PD=PD)


CFGS_F = {(10, 10, 1): cfg_f(10, 10, 3, 16), (14, 14, 1): cfg_f(14, 14, 2, 12), (10, 14, 2): cfg_f(10, 14, 2, 12, 2), (14, 18, 2): cfg_f(14, 18, 1, 12, 2)}


def filter_grad(x, dy, kw, padl, n_wg=3, stride=1):
    """x [B][Tin][H][CI], dy [B][Tout][H][CO] -> (dW [kw][CI][CO], dbias [CO]) computed the way tds_conv_tzf_k + tds_tzf_reduce_k do."""
    B, Tin, H, CI = x.shape
    Tout, CO = dy.shape[1], dy.shape[3]
    g = CFGS_F[(CI, CO, stride)]
    R, GR, SIG, S, C2, NPR, NPT, RF, NFX, NFD, PX, PD = (g[k] for k in ("R", "GR", "SIG", "S", "C2", "NPR", "NPT", "RF", "NFX", "NFD", "PX", "PD"))
    HCI, HCO = H * CI, H * CO
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
                xs[f * PX + HB * CI: (f + 1) * PX] = 1.0
                for piece in range(HB * CI // 4):
                    off = (t0 * SIG - padl + f) * HCI + hb * HB * CI + piece * 4
                    ok = 0 <= off and off + 4 <= Tin * HCI
                    xs[f * PX + piece * 4: f * PX + piece * 4 + 4] = xflat[b, off:off + 4] if ok else 0.0
            for f in range(NFD):
                for piece in range(HB * CO // 4):
                    off = (t0 + f) * HCO + hb * HB * CO + piece * 4
                    ok = 0 <= off and off + 4 <= Tout * HCO
                    ds[f * PD + piece * 4: f * PD + piece * 4 + 4] = dflat[b, off:off + 4] if ok else 0.0
            for wave in range(8):
                rr = np.where(n < R * CO, n // CO, 0)
                co = np.where(n < R * CO, n % CO, 0)
                bBase = rr * PD + (2 * wave + hf) * CO + co
                for i in range(GR):
                    bf = ds[bBase + i * R * PD]
                    for pt in range(NPT):
                        pr = 32 * pt + n
                        s, cp = pr // C2, pr % C2
                        aoff = np.where(pr < NPR, s * PX + (2 * wave + hf) * CI + 2 * cp, HB * CI) + i * SIG * R * PX
                        a0, a1 = xs[aoff], xs[aoff + 1]
                        a0 = np.where(pr > NPR, 0.0, a0)     # rows nobody exports: anything finite
                        a1 = np.where(pr > NPR, 0.0, a1)
                        acc[wave, pt, 0] = mfma_32x32x2(a0, bf, acc[wave, pt, 0])
                        acc[wave, pt, 1] = mfma_32x32x2(a1, bf, acc[wave, pt, 1])
        partial[wg] = acc.sum(axis=0)
    # ---- tds_tzf_reduce_k
    flat = partial.reshape(nblocks, -1)
    nW = kw * CI * CO
    dW = np.zeros(nW)
    db = np.zeros(CO)
    for o in range(nW + CO):
        tot = 0.0
        for r in range(R):
            if o < nW:
                j, rem = divmod(o, CI * CO)
                ci, co = divmod(rem, CO)
                pr, e = (j + SIG * r) * C2 + (ci >> 1), ci & 1
            else:
                co = o - nW
                pr, e = NPR, 0
            pt, row, col = pr >> 5, pr & 31, r * CO + co
            idx = ((pt * 2 + e) * 16 + 4 * (row >> 3) + (row & 3)) * 64 + 32 * ((row >> 2) & 1) + col
            tot += flat[:, idx].sum()
        if o < nW:
            dW[o] = tot
        else:
            db[o - nW] = tot
    return dW.reshape(kw, CI, CO), db


def filter_direct(x, dy, kw, padl, stride=1):
    B, Tin, H, CI = x.shape
    Tout, CO = dy.shape[1], dy.shape[3]
    dW = np.zeros((kw, CI, CO))
    for j in range(kw):
        for t in range(Tout):
            ti = stride * t + j - padl
            if 0 <= ti < Tin:
                dW[j] += np.einsum("bhc,bhd->cd", x[:, ti].astype(np.float64), dy[:, t].astype(np.float64))
    return dW, dy.astype(np.float64).sum(axis=(0, 1, 2))
