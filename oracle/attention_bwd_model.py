"""TEST INFRASTRUCTURE (oracle side).  The INDEX ARITHMETIC of wav2letter_amd/csrc/attention_fused_bwd.hip restated in numpy, float64,
no rounding: what each of its four kernels computes, in the decomposition and the layouts the kernels use --

  * the MFMA C layout of a 32 x 32 tile (lane (li, lh) holds column li and rows (r & 3) + 8 (r >> 2) + 4 lh, r = 0 .. 15),
  * the query side per block of 32 queries: dP^T tiles, the softmax backward, the images dS^T [key][query] and the skewed copy
    dR^T [window row][query] produced by running the forward's LDS skew backwards (scratch[query][w'] <- tile; an entry of window
    block e comes from tile e where wr + li >= 31 and from tile e - 1 below that),
  * the GLOBAL numbering of window rows, wg = w - wOrg with wOrg = (n0 - rlo) - 31 - 32 (NT - 1), in which table block e of query
    block qb is block g = e - qb + NT - 1 for every qb; the E^T image in that numbering (zero outside the table window),
  * the key side: dk = dS^T q, dv = Pd^T dctx, and the table gradient from dR^T q restricted to the query blocks inside the band
    (e = g + qb - (NT - 1) in [0, NT]), reduced over (utterance, head) into the table rows rlo .. rlo + W.

tests/test_attention_bwd_model.py holds it against torch float64 autograd of the attention forward (oracle/transformer_oracle.py's
relative-position rotation; block semantics recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151) on the CPU, so the
geometry of the kernels is pinned where no GPU is needed; the kernels themselves are compared with float64 on the device
(tests/test_gpu_attention.py::test_fused_attention_backward)."""
import numpy as np


def nt_of(T):
    nt = (T + 31) // 32
    for cand in (2, 4, 6):
        if nt <= cand:
            return cand
    raise ValueError("T > 192 has no fused kernel")


def c_layout_rows(lh):
    """rows of a 32 x 32 MFMA C tile held by a lane of half lh, in register order"""
    return np.array([(r & 3) + 8 * (r >> 2) + 4 * lh for r in range(16)])


def fused_backward_model(q, k, v, E, P, dctx, scale, csz):
    """q, k, v, dctx [B][H][T][d]; P [B][H][T][T]; E [2 csz - 1][d] or None.  Returns dq, dk, dv [B][H][T][d], dE [2 csz - 1][d] (or None)."""
    B, H, T, d = q.shape
    NT = nt_of(T)
    TP, GW = 32 * NT, 64 * NT
    n0 = csz - 1
    rlo = max(0, n0 - (T - 1)) if csz else 0
    W = (min(2 * csz - 1, n0 + T) - rlo) if csz else 0
    wOrg = (n0 - rlo) - 31 - 32 * (NT - 1)
    nqb = (T + 31) // 32
    dq = np.zeros_like(q); dk = np.zeros_like(q); dv = np.zeros_like(q)
    dEp = np.zeros((B, H, GW, d))
    Et = None
    if csz:   # attn_bwd_prep_k
        Et = np.zeros((d, GW))
        for wg in range(GW):
            w = wg + wOrg
            if 0 <= w < W:
                Et[:, wg] = E[rlo + w]
    for b in range(B):
        for h in range(H):
            dSt = np.zeros((TP, TP)); Pdt = np.zeros((TP, TP)); dRt = np.full((GW, TP), np.nan)   # NaN = never written
            # ---------------- attn_fused_bwd_q_k: one "wave" per query block
            for qb in range(nqb):
                i0 = 32 * qb
                dS = np.zeros((NT, 2, 16, 32))          # [tile][lh][r][li]
                for li in range(32):
                    i = i0 + li
                    if i >= T:
                        continue
                    dProw = v[b, h] @ dctx[b, h, i]       # dP[i][j] over the keys
                    dot = float(P[b, h, i] @ dProw)
                    row = scale * P[b, h, i] * (dProw - dot)
                    for t in range(NT):
                        for lh in range(2):
                            j = 32 * t + c_layout_rows(lh)
                            ok = j < T
                            dS[t, lh, ok, li] = row[j[ok]]
                for t in range(NT):
                    for lh in range(2):
                        j = 32 * t + c_layout_rows(lh)
                        dSt[j, i0:i0 + 32] = dS[t, lh]
                        for li in range(32):
                            if i0 + li < T:
                                ok = j < T
                                Pdt[j[ok], i0 + li] = P[b, h, i0 + li, j[ok]]
                for li in range(32):
                    if i0 + li < T:
                        dq[b, h, i0 + li] = dSt[:T, i0 + li] @ k[b, h]
                if csz:
                    prev_up = np.zeros((2, 16, 32))
                    for e in range(NT + 1):
                        blk = np.zeros((2, 16, 32)); up = np.zeros((2, 16, 32))
                        if e < NT:
                            scratch = np.full((32, 65), np.nan)   # stale slots: never selected
                            for lh in range(2):
                                jj = c_layout_rows(lh)
                                for li in range(32):
                                    scratch[li, jj - li + 31] = dS[e, lh, :, li]
                            for lh in range(2):
                                wr = c_layout_rows(lh)
                                for li in range(32):
                                    lo = scratch[li, wr]
                                    up[lh, :, li] = scratch[li, 32 + wr]
                                    blk[lh, :, li] = np.where(wr + li >= 31, lo, prev_up[lh, :, li])
                        else:
                            for lh in range(2):
                                wr = c_layout_rows(lh)
                                for li in range(32):
                                    blk[lh, :, li] = np.where(wr + li >= 31, 0.0, prev_up[lh, :, li])
                        prev_up = up
                        g = e - qb + NT - 1
                        assert 0 <= g < 2 * NT
                        for lh in range(2):
                            wr = c_layout_rows(lh)
                            dRt[32 * g + wr, i0:i0 + 32] = blk[lh]
                            assert not np.isnan(blk[lh]).any()
                            for li in range(32):
                                if i0 + li < T:
                                    dq[b, h, i0 + li] += Et[:, 32 * g + wr] @ blk[lh, :, li]
            # ---------------- attn_fused_bwd_kv_k
            dk[b, h] = dSt[:T, :T] @ q[b, h]
            dv[b, h] = Pdt[:T, :T] @ dctx[b, h]
            if csz:
                for g in range(2 * NT):
                    for qb in range(nqb):
                        e = g + qb - (NT - 1)
                        if not 0 <= e <= NT:
                            continue   # (never written for this query block: the kernel does not read it)
                        rows = slice(32 * g, 32 * g + 32)
                        cols = slice(32 * qb, min(32 * qb + 32, T))
                        blk = dRt[rows, cols]
                        assert not np.isnan(blk).any()
                        dEp[b, h, rows] += blk @ q[b, h, cols]
    dE = None
    if csz:   # attn_bwd_de_reduce_k
        dE = np.zeros((2 * csz - 1, d))
        for row in range(2 * csz - 1):
            w = row - rlo
            if 0 <= w < W:
                dE[row] = dEp[:, :, w - wOrg].sum(axis=(0, 1))
    return dq, dk, dv, dE
