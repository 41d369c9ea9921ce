"""fl::Transformer's attention core and the time-axis max pool through the C ABI (wav2letter_amd/csrc/attention.hip):
the strided batched GEMM in every operand orientation the block uses, the softmax with the gathered relative-position
term and its backward, and a whole TR network end to end against oracle/transformer_oracle.py (torch float64 restatement
of recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import refnet

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(got, want):
    want = np.asarray(want, np.float64).reshape(-1)
    got = np.asarray(got, np.float64).reshape(-1)
    assert got.shape == want.shape, (got.shape, want.shape)
    return np.abs(got - want).max() / max(1e-30, np.abs(want).max())


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("B,H,T,d", [(2, 4, 47, 8), (3, 2, 64, 32), (1, 4, 188, 256), (2, 1, 5, 4), (2, 3, 130, 20)])
def test_batched_gemm_head_products(B, H, T, d):
    """QK^T, PV, P^T dC and dS^T Q straight out of frame-major [B][T][H*d] buffers, against float64 einsums"""
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + T)
    Cc = H * d
    q = torch.randn(B, T, Cc, generator=g).cuda()
    k = torch.randn(B, T, Cc, generator=g).cuda()
    P = torch.randn(B, H, T, T, generator=g).cuda()
    qh = q.double().reshape(B, T, H, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, T, H, d).permute(0, 2, 1, 3)
    TC, TT = T * Cc, T * T
    # S = q k^T
    S = torch.full((B, H, T, T), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=T, N=T, K=d, G1=B, G2=H, sam=Cc, sak=1, a1=TC, a2=d, sbk=1, sbn=Cc, b1=TC, b2=d, ldc=T, c1=H * TT, c2=TT)
    assert L.w2l_bgemm_f32(C.byref(D), q.data_ptr(), k.data_ptr(), S.data_ptr(), _stream()) == 0
    assert rel(S.cpu().numpy(), (qh @ kh.transpose(-1, -2)).cpu().numpy()) < TOL
    # ctx = P k  (k in the role of v), written back frame-major
    ctx = torch.full((B, T, Cc), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=T, sak=1, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    assert L.w2l_bgemm_f32(C.byref(D), P.data_ptr(), k.data_ptr(), ctx.data_ptr(), _stream()) == 0
    want = (P.double() @ kh).permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(ctx.cpu().numpy(), want.cpu().numpy()) < TOL
    # dv = P^T q (A read transposed: sam = 1), accumulated onto a previous value
    dv = torch.randn(B, T, Cc, generator=g).cuda()
    dv0 = dv.clone()
    D = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d,
                       accumulate=1)
    assert L.w2l_bgemm_f32(C.byref(D), P.data_ptr(), q.data_ptr(), dv.data_ptr(), _stream()) == 0
    want = dv0.double() + (P.double().transpose(-1, -2) @ qh).permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(dv.cpu().numpy(), want.cpu().numpy()) < TOL


@pytest.mark.parametrize("B,H,T,d", [(2, 4, 47, 8), (3, 2, 64, 32), (1, 4, 188, 256), (2, 1, 5, 4), (2, 3, 130, 20)])
def test_batched_gemm_bf16_operands(B, H, T, d):
    """the mixed-precision attention products: operands rounded to bf16 on the way into LDS, fp32 accumulate and fp32 C.
    Exact restatement = float64 product of the bf16-rounded operands; the accumulate orientation adds onto fp32 unrounded"""
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 77 + T)
    Cc = H * d
    q = torch.randn(B, T, Cc, generator=g).cuda()
    k = torch.randn(B, T, Cc, generator=g).cuda()
    P = torch.randn(B, H, T, T, generator=g).cuda()
    r = lambda x: x.bfloat16().double()
    qh = r(q).reshape(B, T, H, d).permute(0, 2, 1, 3)
    kh = r(k).reshape(B, T, H, d).permute(0, 2, 1, 3)
    TC, TT = T * Cc, T * T
    S = torch.full((B, H, T, T), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=T, N=T, K=d, G1=B, G2=H, sam=Cc, sak=1, a1=TC, a2=d, sbk=1, sbn=Cc, b1=TC, b2=d, ldc=T, c1=H * TT, c2=TT)
    assert L.w2l_bgemm_bf16(C.byref(D), q.data_ptr(), k.data_ptr(), S.data_ptr(), _stream()) == 0
    assert rel(S.cpu().numpy(), (qh @ kh.transpose(-1, -2)).cpu().numpy()) < 2e-5
    ctx = torch.full((B, T, Cc), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=T, sak=1, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    assert L.w2l_bgemm_bf16(C.byref(D), P.data_ptr(), k.data_ptr(), ctx.data_ptr(), _stream()) == 0
    want = (r(P) @ kh).permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(ctx.cpu().numpy(), want.cpu().numpy()) < 2e-5
    dv = torch.randn(B, T, Cc, generator=g).cuda()
    dv0 = dv.clone()
    D = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d,
                       accumulate=1)
    assert L.w2l_bgemm_bf16(C.byref(D), P.data_ptr(), q.data_ptr(), dv.data_ptr(), _stream()) == 0
    want = dv0.double() + (r(P).transpose(-1, -2) @ qh).permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(dv.cpu().numpy(), want.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("B,H,T,d,csz", [(2, 4, 23, 8, 30), (2, 2, 40, 16, 7), (1, 4, 188, 64, 460), (3, 1, 1, 4, 2)])
def test_relative_position_products_and_softmax(B, H, T, d, csz):
    """R = Qflat E_win^T, softmax(scale (S + skew(R))) and its backward (dS and the skewed dR) against the float64
    restatement of multiheadAttention's score path"""
    from oracle import transformer_oracle as TO
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T * 10 + csz)
    Cc = H * d
    q = torch.randn(B, T, Cc, generator=g)
    S = torch.randn(B, H, T, T, generator=g)
    E = torch.randn(2 * csz - 1, d, generator=g) * 0.5
    dP = torch.randn(B, H, T, T, generator=g)
    n0 = csz - 1
    rlo = max(0, n0 - (T - 1))
    W = min(2 * csz - 1, n0 + T) - rlo
    ldr = (W + 3) // 4 * 4
    scale = 1.0 / np.sqrt(d)
    qd, Sd, Ed = q.cuda(), S.cuda(), E.cuda()
    R = torch.full((B * T * H, ldr), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=B * T * H, N=W, K=d, G1=1, G2=1, sam=d, sak=1, sbk=1, sbn=d, ldc=ldr)
    assert L.w2l_bgemm_f32(C.byref(D), qd.data_ptr(), Ed[rlo:].data_ptr(), R.data_ptr(), _stream()) == 0
    Pd = Sd.clone()
    assert L.w2l_attn_softmax_forward(Pd.data_ptr(), R.data_ptr(), None, B, H, T, ldr, rlo, W, n0, scale, _stream()) == 0
    # oracle
    S64 = S.double().requires_grad_(True)
    q64 = q.double().requires_grad_(True)
    E64 = E.double().requires_grad_(True)
    qh = q64.reshape(B, T, H, d).permute(0, 2, 1, 3)
    rot = TO.relative_position_rotate(qh @ E64.t())
    n = E.shape[0] // 2
    Pref = torch.softmax((S64 + rot[..., n:n + T]) * scale, dim=-1)
    assert rel(Pd.cpu().numpy(), Pref.detach().numpy()) < TOL
    Pref.backward(dP.double())
    dS = dP.cuda().clone()
    dR = torch.full((B * T * H, ldr), float("nan"), device="cuda")
    assert L.w2l_attn_softmax_backward(Pd.data_ptr(), dS.data_ptr(), dR.data_ptr(), B, H, T, ldr, rlo, W, n0, scale, _stream()) == 0
    assert rel(dS.cpu().numpy(), S64.grad.numpy()) < TOL
    assert torch.isfinite(dR).all()
    # dq (relative part) = dR E_win, dE_win = dR^T Qflat
    dq = torch.zeros(B, T, Cc, device="cuda")
    D = _lib.BgemmDesc(M=B * T * H, N=d, K=W, G1=1, G2=1, sam=ldr, sak=1, sbk=d, sbn=1, ldc=d, accumulate=1)
    assert L.w2l_bgemm_f32(C.byref(D), dR.data_ptr(), Ed[rlo:].data_ptr(), dq.data_ptr(), _stream()) == 0
    assert rel(dq.cpu().numpy(), q64.grad.numpy()) < TOL
    dEp = torch.full((B, W, d), float("nan"), device="cuda")
    D = _lib.BgemmDesc(M=W, N=d, K=T * H, G1=B, G2=1, sam=1, sak=ldr, a1=T * H * ldr, sbk=d, sbn=1, b1=T * Cc, ldc=d, c1=W * d)
    assert L.w2l_bgemm_f32(C.byref(D), dR.data_ptr(), qd.data_ptr(), dEp.data_ptr(), _stream()) == 0
    dE = torch.zeros(2 * csz - 1, d, device="cuda")
    assert L.w2l_colsum(dEp.data_ptr(), dE[rlo:].data_ptr(), B, W * d, _stream()) == 0
    assert rel(dE.cpu().numpy(), E64.grad.numpy()) < TOL
    # no position term: plain softmax
    P0 = Sd.clone()
    assert L.w2l_attn_softmax_forward(P0.data_ptr(), None, None, B, H, T, 0, 0, 0, 0, scale, _stream()) == 0
    assert rel(P0.cpu().numpy(), torch.softmax(S.double() * scale, -1).numpy()) < TOL


@pytest.mark.parametrize("B,H,T,d,csz,p,ragged", [(2, 4, 188, 256, 460, 0.0, False), (2, 4, 188, 256, 460, 0.2, True),
                                                   (1, 2, 64, 32, 7, 0.0, False), (2, 3, 130, 32, 460, 0.3, False),
                                                   (2, 2, 33, 32, 0, 0.0, True), (3, 1, 1, 32, 2, 0.0, False),
                                                   (1, 2, 190, 32, 40, 0.1, True)])
def test_fused_attention_forward(B, H, T, d, csz, p, ragged):
    """w2l_attn_fused_forward (scores + position term + padding mask + softmax + dropout + P V in one launch) against
    (a) the float64 restatement of TransformerCPC.cpp:117-151's score path on the SAME bf16-rounded operands -- only the fp32
    accumulation differs: 2e-5 of the largest magnitude for P, and ctx from the kernel's own P --, (b) the unfused launch
    sequence (bit-identical dropout pattern: the hash runs over the same [B][H][T][T] index), incl. the recipe geometry
    1024 / 4 heads / 919-row table / 188 frames, a ragged batch, T not a multiple of 4, one frame, no table"""
    from oracle import transformer_oracle as TO
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T * 7 + d + csz)
    Cc = H * d
    q = torch.randn(B, T, Cc, generator=g).cuda()
    k = torch.randn(B, T, Cc, generator=g).cuda()
    v = torch.randn(B, T, Cc, generator=g).cuda()
    E = (torch.randn(max(1, 2 * csz - 1), d, generator=g) * 0.5).cuda()
    n0 = csz - 1
    rlo = max(0, n0 - (T - 1)) if csz else 0
    W = (min(2 * csz - 1, n0 + T) - rlo) if csz else 0
    scale = 1.0 / np.sqrt(d)
    keyLen = None
    if ragged:
        keyLen = torch.tensor([T] + [max(1, (T * (3 + b)) // (5 + b)) for b in range(1, B)], dtype=torch.int32).cuda()
    seed, sid = 4321, 5
    P = torch.full((B, H, T, T), float("nan"), device="cuda")
    Pd = torch.full((B, H, T, T), float("nan"), device="cuda")
    ctx = torch.full((B, T, Cc), float("nan"), device="cuda")
    D = _lib.AttnFusedDesc(B=B, H=H, T=T, d=d, ld=Cc, ldc=Cc, W=W, n0=n0, rlo=rlo, scale=scale, dropP=p, dropSeed=seed, dropStream=sid)
    st = L.w2l_attn_fused_forward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None,
                                  keyLen.data_ptr() if ragged else None, P.data_ptr(), Pd.data_ptr() if p > 0 else None,
                                  ctx.data_ptr(), _stream())
    assert st == 0
    # (a) float64 on bf16-rounded operands
    r = lambda x: x.bfloat16().double()
    qh = r(q).reshape(B, T, H, d).permute(0, 2, 1, 3)
    kh = r(k).reshape(B, T, H, d).permute(0, 2, 1, 3)
    vh = r(v).reshape(B, T, H, d).permute(0, 2, 1, 3)
    S = qh @ kh.transpose(-1, -2)
    if csz:
        rot = TO.relative_position_rotate((qh @ r(E).t()).cpu()).cuda()
        n = E.shape[0] // 2
        S = S + rot[..., n:n + T]
    S = S * scale
    if ragged:
        jj = torch.arange(T, device="cuda")
        S = S.masked_fill(jj[None, None, None, :] >= keyLen.long()[:, None, None, None], float("-inf"))
    Pref = torch.softmax(S, dim=-1)
    assert rel(P.cpu().numpy(), Pref.cpu().numpy()) < 2e-5
    Puse = Pd if p > 0 else P
    cref = (r(Puse) @ vh).permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(ctx.cpu().numpy(), cref.cpu().numpy()) < 2e-5
    # (b) the unfused sequence
    TC, TT = T * Cc, T * T
    S2 = torch.full((B, H, T, T), float("nan"), device="cuda")
    G = _lib.BgemmDesc(M=T, N=T, K=d, G1=B, G2=H, sam=Cc, sak=1, a1=TC, a2=d, sbk=1, sbn=Cc, b1=TC, b2=d, ldc=T, c1=H * TT, c2=TT)
    assert L.w2l_bgemm_bf16(C.byref(G), q.data_ptr(), k.data_ptr(), S2.data_ptr(), _stream()) == 0
    ldr = (W + 3) // 4 * 4
    R = None
    if csz:
        R = torch.full((B * T * H, ldr), float("nan"), device="cuda")
        G = _lib.BgemmDesc(M=B * T * H, N=W, K=d, G1=1, G2=1, sam=d, sak=1, sbk=1, sbn=d, ldc=ldr)
        assert L.w2l_bgemm_bf16(C.byref(G), q.data_ptr(), E[rlo:].data_ptr(), R.data_ptr(), _stream()) == 0
    assert L.w2l_attn_softmax_forward(S2.data_ptr(), R.data_ptr() if csz else None, keyLen.data_ptr() if ragged else None,
                                      B, H, T, ldr, rlo, W, n0, scale, _stream()) == 0
    assert rel(P.cpu().numpy(), S2.cpu().numpy()) < 2e-6
    if p > 0:
        Pd2 = torch.empty_like(S2)
        assert L.w2l_dropout_copy(Pd2.data_ptr(), P.data_ptr(), B * H * TT, p, seed, sid, _stream()) == 0
        assert torch.equal(Pd2, Pd)                                       # same keep pattern, same scaling
        assert 0.5 * p < float((Pd == 0).float().mean()) - float((P == 0).float().mean()) < 1.5 * p


@pytest.mark.parametrize("B,H,T,d,csz,p,ragged", [(2, 4, 188, 256, 460, 0.0, False), (2, 4, 188, 256, 460, 0.2, True),
                                                   (1, 2, 64, 32, 7, 0.0, False), (2, 3, 130, 32, 460, 0.3, False),
                                                   (2, 2, 33, 32, 0, 0.0, True), (3, 1, 1, 32, 2, 0.0, False),
                                                   (1, 2, 190, 32, 40, 0.1, True), (2, 2, 96, 256, 3, 0.25, False),
                                                   (2, 4, 37, 256, 460, 0.0, False)])
def test_fused_attention_backward(B, H, T, d, csz, p, ragged):
    """w2l_attn_fused_backward (dP, dropout mask, softmax backward, dq incl. the position term, dk, dv, table gradient in four
    launches) on the P of the fused forward, against (a) the float64 restatement of the gradient of TransformerCPC.cpp:117-151 on
    the SAME bf16-rounded operands: the kernel's dS image (read from the documented head of the workspace) at bf16 resolution
    against the float64 dS, its Pd image bit for bit, and dq / dk / dv / dE from the kernel's own dS at 2e-5 of the largest
    magnitude (only fp32 accumulation differs); (b) the unfused launch sequence of host/net.cpp at the bar a flipped bf16
    rounding of one dS entry allows.  Same geometries as the forward test plus a short table (band narrower than T)."""
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T * 11 + d + csz)
    Cc = H * d
    q = torch.randn(B, T, Cc, generator=g).cuda()
    k = torch.randn(B, T, Cc, generator=g).cuda()
    v = torch.randn(B, T, Cc, generator=g).cuda()
    dctx = torch.randn(B, T, Cc, generator=g).cuda()
    E = (torch.randn(max(1, 2 * csz - 1), d, generator=g) * 0.5).cuda()
    n0 = csz - 1
    rlo = max(0, n0 - (T - 1)) if csz else 0
    W = (min(2 * csz - 1, n0 + T) - rlo) if csz else 0
    scale = 1.0 / np.sqrt(d)
    keyLen = None
    if ragged:
        keyLen = torch.tensor([T] + [max(1, (T * (3 + b)) // (5 + b)) for b in range(1, B)], dtype=torch.int32).cuda()
    seed, sid = 977, 3
    P = torch.full((B, H, T, T), float("nan"), device="cuda")
    Pd = torch.full((B, H, T, T), float("nan"), device="cuda")
    ctx = torch.full((B, T, Cc), float("nan"), device="cuda")
    D = _lib.AttnFusedDesc(B=B, H=H, T=T, d=d, ld=Cc, ldc=Cc, W=W, n0=n0, rlo=rlo, scale=scale, dropP=p, dropSeed=seed, dropStream=sid)
    assert L.w2l_attn_fused_forward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None,
                                    keyLen.data_ptr() if ragged else None, P.data_ptr(), Pd.data_ptr() if p > 0 else None,
                                    ctx.data_ptr(), _stream()) == 0
    if p == 0:
        Pd = P
    nws = L.w2l_attn_fused_backward_workspace(C.byref(D), 1 if csz else 0)
    assert nws > 0
    ws = torch.full((nws,), 0xff, dtype=torch.uint8, device="cuda")     # (0xffff: a bf16 NaN in every unwritten slot)
    dq = torch.full((B, T, Cc), float("nan"), device="cuda")
    dk = torch.full((B, T, Cc), float("nan"), device="cuda")
    dv = torch.full((B, T, Cc), float("nan"), device="cuda")
    dE = torch.full((max(1, 2 * csz - 1), d), float("nan"), device="cuda")
    st = L.w2l_attn_fused_backward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None, P.data_ptr(),
                                   dctx.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dE.data_ptr() if csz else None,
                                   ws.data_ptr(), nws, _stream())
    assert st == 0
    assert L.w2l_attn_fused_backward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None, P.data_ptr(),
                                     dctx.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dE.data_ptr() if csz else None,
                                     ws.data_ptr(), nws - 1, _stream()) != 0      # a short workspace is refused
    for t_ in (dq, dk, dv):
        assert torch.isfinite(t_).all()
    # the same results as bf16 images written by the kernels (w2l_attn_fused_backward_images, no fp32 copies): bit for bit the
    # rounding of the fp32 results, nothing outside the matrix touched; likewise ctx of the forward call
    M = B * T
    ldR, ldT = (Cc + 63) // 64 * 64 + 64, (M + 63) // 64 * 64
    def sink():
        rows = torch.full((M, ldR), 7.0, dtype=torch.bfloat16, device="cuda")
        trans = torch.full((Cc, ldT), 7.0, dtype=torch.bfloat16, device="cuda")
        return rows, trans, _lib.Bf16ImageSink(rowMajor=rows.data_ptr(), ldRows=ldR, transposed=trans.data_ptr(), ldTrans=ldT)
    sinks = [sink() for _ in range(4)]
    dE_i = torch.full_like(dE, float("nan"))
    assert L.w2l_attn_fused_backward_images(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None, P.data_ptr(),
                                            dctx.data_ptr(), None, None, None, C.byref(sinks[0][2]), C.byref(sinks[1][2]), C.byref(sinks[2][2]),
                                            dE_i.data_ptr() if csz else None, ws.data_ptr(), nws, _stream()) == 0
    P_i, Pd_i = torch.empty_like(P), torch.empty_like(P)
    assert L.w2l_attn_fused_forward_images(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None,
                                           keyLen.data_ptr() if ragged else None, P_i.data_ptr(), Pd_i.data_ptr() if p > 0 else None,
                                           None, C.byref(sinks[3][2]), _stream()) == 0
    assert torch.equal(P_i, P)
    for (rows, trans, _), ref32 in zip(sinks, (dq, dk, dv, ctx)):
        want = ref32.reshape(M, Cc).bfloat16()
        assert torch.equal(rows[:, :Cc], want) and torch.equal(trans[:, :M], want.t())
        assert bool((rows[:, Cc:] == 7.0).all()) and bool((trans[:, M:] == 7.0).all())
    if csz:
        assert torch.equal(dE_i, dE)
    assert L.w2l_attn_fused_backward_images(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr() if csz else None, P.data_ptr(),
                                            dctx.data_ptr(), None, dk.data_ptr(), dv.data_ptr(), None, None, None,
                                            dE_i.data_ptr() if csz else None, ws.data_ptr(), nws, _stream()) != 0   # neither dq nor its images
    # (a) float64 on bf16-rounded operands
    r = lambda x: x.bfloat16().double()
    heads = lambda x: x.reshape(B, T, H, d).permute(0, 2, 1, 3)
    qh, kh, vh, dch = heads(r(q)), heads(r(k)), heads(r(v)), heads(r(dctx))
    ks = 1.0 / (1.0 - p)
    dP = dch @ vh.transpose(-1, -2)
    if p > 0:
        dP = dP * (Pd != 0).double() * ks
    P64 = P.double()
    dS64 = scale * P64 * (dP - (P64 * dP).sum(-1, keepdim=True))
    nt = (T + 31) // 32
    TP = 32 * (2 if nt <= 2 else 4 if nt <= 4 else 6)
    img = ws[:2 * B * H * TP * TP * 2].view(torch.bfloat16).view(2, B, H, TP, TP)
    dSk = img[0, :, :, :T, :T].transpose(-1, -2).double()          # the kernel's dS, [query][key]
    assert rel(dSk.cpu().numpy(), dS64.cpu().numpy()) < 4.2e-3     # bf16: 2^-8 of the largest entry
    assert torch.equal(img[1, :, :, :T, :T].transpose(-1, -2), Pd.bfloat16())
    if TP > T:
        assert float(img[0, :, :, T:, :T].float().abs().max()) == 0.0 and float(img[1, :, :, T:, :T].float().abs().max()) == 0.0   # padded key rows
    dq_ref = dSk @ kh
    dk_ref = dSk.transpose(-1, -2) @ qh
    dv_ref = r(Pd).transpose(-1, -2) @ dch
    dE_ref = torch.zeros(max(1, 2 * csz - 1), d, dtype=torch.float64, device="cuda")
    if csz:
        Er = r(E)
        for delta in range(-(T - 1), T):
            w = delta + n0
            if not 0 <= w < 2 * csz - 1:
                continue
            diag = dSk.diagonal(offset=delta, dim1=-2, dim2=-1)    # [B][H][n]: entries (i, i + delta)
            lo = max(0, -delta)
            n = diag.shape[-1]
            dq_ref[:, :, lo:lo + n] += diag[..., None] * Er[w]
            dE_ref[w] += (diag[..., None] * qh[:, :, lo:lo + n]).sum((0, 1, 2))
    unheads = lambda x: x.permute(0, 2, 1, 3).reshape(B, T, Cc)
    assert rel(dq.cpu().numpy(), unheads(dq_ref).cpu().numpy()) < 2e-5
    assert rel(dk.cpu().numpy(), unheads(dk_ref).cpu().numpy()) < 2e-5
    assert rel(dv.cpu().numpy(), unheads(dv_ref).cpu().numpy()) < 2e-5
    if csz:
        assert torch.isfinite(dE).all()
        assert rel(dE.cpu().numpy(), dE_ref.cpu().numpy()) < 2e-5
        if rlo > 0:
            assert float(dE[:rlo].abs().max()) == 0.0             # rows no frame pair reaches
    # (b) the unfused sequence (host/net.cpp before this kernel)
    TC, TT = T * Cc, T * T
    dS2 = torch.full((B, H, T, T), float("nan"), device="cuda")
    G = _lib.BgemmDesc(M=T, N=T, K=d, G1=B, G2=H, sam=Cc, sak=1, a1=TC, a2=d, sbk=1, sbn=Cc, b1=TC, b2=d, ldc=T, c1=H * TT, c2=TT)
    assert L.w2l_bgemm_bf16(C.byref(G), dctx.data_ptr(), v.data_ptr(), dS2.data_ptr(), _stream()) == 0
    dv2 = torch.full((B, T, Cc), float("nan"), device="cuda")
    G = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    assert L.w2l_bgemm_bf16(C.byref(G), Pd.data_ptr(), dctx.data_ptr(), dv2.data_ptr(), _stream()) == 0
    if p > 0:
        assert L.w2l_dropout_inplace(dS2.data_ptr(), B * H * TT, p, seed, sid, _stream()) == 0
    ldr = (W + 3) // 4 * 4
    dR = torch.full((B * T * H, max(ldr, 4)), float("nan"), device="cuda")
    assert L.w2l_attn_softmax_backward(P.data_ptr(), dS2.data_ptr(), dR.data_ptr() if csz else None, B, H, T, ldr, rlo, W, n0, scale,
                                       _stream()) == 0
    assert rel(dSk.cpu().numpy(), dS2.cpu().numpy()) < 4.2e-3
    dq2 = torch.full((B, T, Cc), float("nan"), device="cuda")
    G = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=T, sak=1, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    assert L.w2l_bgemm_bf16(C.byref(G), dS2.data_ptr(), k.data_ptr(), dq2.data_ptr(), _stream()) == 0
    dk2 = torch.full((B, T, Cc), float("nan"), device="cuda")
    G = _lib.BgemmDesc(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    assert L.w2l_bgemm_bf16(C.byref(G), dS2.data_ptr(), q.data_ptr(), dk2.data_ptr(), _stream()) == 0
    if csz:
        G = _lib.BgemmDesc(M=B * T * H, N=d, K=W, G1=1, G2=1, sam=ldr, sak=1, sbk=d, sbn=1, ldc=d, accumulate=1)
        assert L.w2l_bgemm_bf16(C.byref(G), dR.data_ptr(), E[rlo:].data_ptr(), dq2.data_ptr(), _stream()) == 0
        dEp = torch.full((B, W, d), float("nan"), device="cuda")
        G = _lib.BgemmDesc(M=W, N=d, K=T * H, G1=B, G2=1, sam=1, sak=ldr, a1=T * H * ldr, sbk=d, sbn=1, b1=T * Cc, ldc=d, c1=W * d)
        assert L.w2l_bgemm_bf16(C.byref(G), dR.data_ptr(), q.data_ptr(), dEp.data_ptr(), _stream()) == 0
        dE2 = torch.zeros(2 * csz - 1, d, device="cuda")
        assert L.w2l_colsum(dEp.data_ptr(), dE2[rlo:].data_ptr(), B, W * d, _stream()) == 0
        assert rel(dE.cpu().numpy(), dE2.cpu().numpy()) < 1e-3
    assert rel(dv.cpu().numpy(), dv2.cpu().numpy()) < 2e-5
    assert rel(dq.cpu().numpy(), dq2.cpu().numpy()) < 1e-3
    assert rel(dk.cpu().numpy(), dk2.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("B,H,T,d,csz", [(2, 2, 37, 32, 460), (1, 2, 130, 32, 460), (2, 4, 188, 256, 460)])
def test_fused_attention_whole_table_window(B, H, T, d, csz):
    """w2l_attn_fused_desc's window may be WIDER than the rows T frames reach ("entries outside count as 0"): with rlo = 0 and
    W = 2 csz - 1 (the whole table) the fused forward and backward give bit for bit what the clipped window gives, and the table
    gradient is zero on the rows no frame pair reaches (the reduce kernel must not look for partial sums there)."""
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T + d)
    Cc = H * d
    q, k, v, dctx = (torch.randn(B, T, Cc, generator=g).cuda() for _ in range(4))
    E = (torch.randn(2 * csz - 1, d, generator=g) * 0.5).cuda()
    n0 = csz - 1
    out = []
    for (rlo, W) in [(max(0, n0 - (T - 1)), min(2 * csz - 1, n0 + T) - max(0, n0 - (T - 1))), (0, 2 * csz - 1)]:
        D = _lib.AttnFusedDesc(B=B, H=H, T=T, d=d, ld=Cc, ldc=Cc, W=W, n0=n0, rlo=rlo, scale=d ** -0.5, dropP=0.1, dropSeed=5, dropStream=2)
        P = torch.full((B, H, T, T), float("nan"), device="cuda")
        Pd = torch.full((B, H, T, T), float("nan"), device="cuda")
        ctx = torch.full((B, T, Cc), float("nan"), device="cuda")
        assert L.w2l_attn_fused_forward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr(), None, P.data_ptr(), Pd.data_ptr(),
                                        ctx.data_ptr(), _stream()) == 0
        nws = L.w2l_attn_fused_backward_workspace(C.byref(D), 1)
        ws = torch.full((nws,), 0xff, dtype=torch.uint8, device="cuda")
        dq, dk, dv = (torch.full((B, T, Cc), float("nan"), device="cuda") for _ in range(3))
        dE = torch.full((2 * csz - 1, d), float("nan"), device="cuda")
        assert L.w2l_attn_fused_backward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr(), P.data_ptr(), dctx.data_ptr(),
                                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dE.data_ptr(), ws.data_ptr(), nws, _stream()) == 0
        torch.cuda.synchronize()
        out.append((P, Pd, ctx, dq, dk, dv, dE))
    for a, b in zip(*out):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    reach = torch.zeros(2 * csz - 1, dtype=torch.bool)
    reach[max(0, n0 - (T - 1)):min(2 * csz - 1, n0 + T)] = True
    assert float(out[1][6][~reach.cuda()].abs().max()) == 0.0 and float(out[1][6][reach.cuda()].abs().min()) > 0.0


@pytest.mark.parametrize("B,H,T,d,csz", [(2, 4, 188, 256, 460), (3, 2, 50, 32, 30), (2, 3, 130, 20, 460), (1, 1, 7, 8, 3), (2, 2, 64, 16, 10)])
def test_banded_position_products_bf16(B, H, T, d, csz):
    """the two relative-position products of the attention backward on the bf16 batched GEMM with the BAND of the skewed score
    gradient declared (w2l_bgemm_desc::bandMode): dq += dR E (rows (b, i, h), k = w) and dE = dR^T q (rows w, k = (i, h)) skip the
    K tiles where dR is zero by construction -- bit-identical to the dense launches"""
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T + csz)
    Cc = H * d
    n0 = csz - 1
    rlo = max(0, n0 - (T - 1))
    W = min(2 * csz - 1, n0 + T) - rlo
    ldr = (W + 3) // 4 * 4
    P = torch.softmax(torch.randn(B, H, T, T, generator=g), -1).cuda()
    dS = torch.randn(B, H, T, T, generator=g).cuda()
    dR = torch.full((B * T * H, ldr), float("nan"), device="cuda")
    assert L.w2l_attn_softmax_backward(P.data_ptr(), dS.data_ptr(), dR.data_ptr(), B, H, T, ldr, rlo, W, n0, d ** -0.5, _stream()) == 0
    E = (torch.randn(2 * csz - 1, d, generator=g) * 0.5).cuda()
    q = torch.randn(B, T, Cc, generator=g).cuda()
    outs = {}
    for band in (0, 1):
        dq = torch.ones(B, T, Cc, device="cuda")
        D = _lib.BgemmDesc(M=B * T * H, N=d, K=W, G1=1, G2=1, sam=ldr, sak=1, sbk=d, sbn=1, ldc=d, accumulate=1,
                           bandMode=1 if band else 0, bandT=T, bandH=H, bandOff=n0 - rlo)
        assert L.w2l_bgemm_bf16(C.byref(D), dR.data_ptr(), E[rlo:].data_ptr(), dq.data_ptr(), _stream()) == 0
        dEp = torch.full((B, W, d), float("nan"), device="cuda")
        D = _lib.BgemmDesc(M=W, N=d, K=T * H, G1=B, G2=1, sam=1, sak=ldr, a1=T * H * ldr, sbk=d, sbn=1, b1=T * Cc, ldc=d, c1=W * d,
                           bandMode=2 if band else 0, bandT=T, bandH=H, bandOff=n0 - rlo)
        assert L.w2l_bgemm_bf16(C.byref(D), dR.data_ptr(), q.data_ptr(), dEp.data_ptr(), _stream()) == 0
        outs[band] = (dq, dEp)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    bad = _lib.BgemmDesc(M=4, N=4, K=4, G1=1, G2=1, sam=4, sak=1, sbk=4, sbn=1, ldc=4, bandMode=3, bandT=1, bandH=1)
    assert L.w2l_bgemm_bf16(C.byref(bad), dR.data_ptr(), E.data_ptr(), q.data_ptr(), _stream()) != 0


@pytest.mark.parametrize("B,T,F,w,stride", [(2, 21, 32, 1, 2), (3, 20, 12, 2, 2), (2, 17, 8, 3, 2), (1, 9, 4, 3, 1)])
def test_time_max_pool(B, T, F, w, stride):
    from wav2letter_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, F, generator=g)
    To = (T - w) // stride + 1
    xr = x.double().permute(0, 2, 1).requires_grad_(True)                 # [B][F][T]
    yr = torch.nn.functional.max_pool1d(xr, w, stride)
    dy = torch.randn(B, To, F, generator=g)
    yr.backward(dy.double().permute(0, 2, 1))
    xd, dyd = x.cuda(), dy.cuda()
    y = torch.empty(B, To, F, device="cuda")
    dx = torch.full((B, T, F), float("nan"), device="cuda")
    assert L.w2l_pool_time_forward(xd.data_ptr(), y.data_ptr(), B, T, F, w, stride, _stream()) == 0
    assert L.w2l_pool_time_backward(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), B, T, F, w, stride, _stream()) == 0
    assert torch.equal(y.cpu().double(), yr.detach().permute(0, 2, 1))
    assert rel(dx.cpu().numpy(), xr.grad.permute(0, 2, 1).numpy()) < 1e-6


@pytest.mark.parametrize("Tin,Tk", [(1500, 188), (44, 22), (37, 5), (64, 64), (10, 3)])
def test_padding_mask_key_lengths_and_masked_softmax(Tin, Tk):
    """valid keys per utterance from the batch's input sizes (forwardSequentialModuleWithPadMask + af::resize, restated in
    oracle/transformer_oracle.key_lengths) bit-exact, and the softmax with the padded keys at probability 0"""
    from oracle import transformer_oracle as TO
    from wav2letter_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(Tin + Tk)
    B, H = 7, 2
    sizes = np.concatenate([[16000.0 * 9.3], rng.uniform(0.05, 1.0, size=B - 1) * 16000.0 * 9.3]).astype(np.float32)
    kl = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    assert L.w2l_attn_key_lengths(torch.tensor(sizes).cuda().data_ptr(), B, Tin, Tk, kl.data_ptr(), _stream()) == 0
    want = TO.key_lengths(sizes, Tin, Tk)
    assert (kl.cpu().numpy() == want).all(), (kl.cpu().numpy(), want)
    assert want[0] == Tk and want.min() >= 1
    S = torch.randn(B, H, Tk, Tk, generator=torch.Generator().manual_seed(1))
    P = S.cuda().clone()
    assert L.w2l_attn_softmax_forward(P.data_ptr(), None, kl.data_ptr(), B, H, Tk, 0, 0, 0, 0, 0.5, _stream()) == 0
    pad = torch.arange(Tk)[None, :] >= torch.tensor(want)[:, None]
    ref = torch.softmax((S.double() * 0.5).masked_fill(pad[:, None, None, :], float("-inf")), -1)
    assert rel(P.cpu().numpy(), ref.numpy()) < TOL
    assert (P.cpu()[pad[:, None, None, :].expand_as(P)] == 0).all()


@pytest.mark.parametrize("seed", [1, 2])
def test_fused_attention_random_geometries(seed):
    """the two tests above over random batch / head / frame counts (T = 1 ... 200, on and off the 4- and 16-frame tile edges), both
    head widths the fused kernels hold, tables shorter and longer than the sequence, with and without dropout and ragged key lengths"""
    rng = np.random.default_rng(700 + seed)
    for c in range(10):
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        T = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 188, 191, 192]))   # (the fused kernels hold T <= 192)
        d = int(rng.choice([32, 256]))
        csz = int(rng.choice([0, 1, 2, 7, 40, 200, 460]))
        p = float(rng.choice([0.0, 0.2]))
        ragged = bool(rng.integers(0, 2))
        try:
            test_fused_attention_forward(B, H, T, d, csz, p, ragged)
            test_fused_attention_backward(B, H, T, d, csz, p, ragged)
        except AssertionError as e:
            raise AssertionError(f"case {c}: B={B} H={H} T={T} d={d} csz={csz} p={p} ragged={ragged}: {e}") from e
