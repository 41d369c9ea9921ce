"""Parity of the HIP network operators (GEMM engine, implicit-GEMM time
convolution, fused residual/dropout/LayerNorm, GLU, SGD) against the CPU oracle.
Tolerance: 1e-4 relative to the largest reference magnitude (fp32)."""
import time

import numpy as np
import pytest
import torch

from oracle import refnet

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.fixture
def probe():
    """kernel-variant tests run on libw2l_hip_probe.so: the only build that reads the W2L_* switches"""
    from wav2letter_amd import _lib
    with _lib.use_probe() as L:
        yield L


def rel(got, want):
    want = np.asarray(want, np.float64)
    got = np.asarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return np.abs(got - want).max() / max(1e-30, np.abs(want).max())


def to_fm(x):  # reference [B][C][H][T] -> frame-major [B][T][H][C]
    return np.ascontiguousarray(x.transpose(0, 3, 2, 1))


def from_fm(y):
    return np.ascontiguousarray(y.transpose(0, 3, 2, 1))


def w_to_dev(w):  # reference [Cout][Cin][kw] -> [kw][Cin][Cout]
    return np.ascontiguousarray(w.transpose(2, 1, 0))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (128, 128, 32), (130, 250, 70), (257, 129, 33),
                                    (300, 9998, 64), (64, 10, 210)])
@pytest.mark.parametrize("akc,bkc", [(True, False), (True, True), (False, False), (False, True)])
def test_gemm_all_layouts(oracle, M, N, K, akc, bkc):
    from wav2letter_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    Bm = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    want = A.astype(np.float64) @ Bm.astype(np.float64) + bias
    Ad = dev(A if akc else A.T)
    Bd = dev(Bm.T if bkc else Bm)
    got = ops.gemm(Ad, Bd, akc, bkc, dev(bias))
    assert rel(got, want) < TOL
    got = ops.gemm(Ad, Bd, akc, bkc, dev(bias), relu=True)
    assert rel(got, np.maximum(want, 0)) < TOL
    if K >= 64:
        got = ops.gemm(Ad, Bd, akc, bkc, splitk=2)
        assert rel(got, want - bias) < TOL


@pytest.mark.parametrize("M,N,K", [(256, 384, 2000),      # 6 tiles < 512 slots: pure stream-K, ranges span two tiles
                                    (6016, 1440, 416),    # 564 tiles: one data-parallel round + stream-K tail
                                    (1500, 700, 4096),    # 72 tiles, long K: many partial slabs per tile
                                    (129, 129, 8192)])    # ragged edge tiles through the slab fix-up
@pytest.mark.parametrize("akc,bkc", [(True, False), (False, False), (True, True)])
def test_gemm_stream_k_schedule(M, N, K, akc, bkc):
    """the hybrid data-parallel + stream-K schedule (partial slabs + ordered fix-up) against a
    float64 product, bias + ReLU through the fix-up epilogue, and run-to-run determinism"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = (A.double() @ Bm.double() + bias.double()).numpy()
    Ad = (A if akc else A.T.contiguous()).cuda()
    Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
    got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    assert rel(got, want) < TOL
    got2 = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    assert torch.equal(got, got2)
    got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda(), relu=True)
    assert rel(got, np.maximum(want, 0)) < TOL


@pytest.mark.parametrize("M,N,K", [(4, 4, 32),             # one clamped tile, one K step
                                    (260, 388, 96),        # ragged edge tiles in both directions
                                    (1028, 2052, 1440),    # 153 tiles: pure stream-K ranges crossing tiles
                                    (6016, 4320, 1440),    # 1598 tiles: 3 persistent rounds + stream-K tail
                                    (640, 136, 24000),     # dW-like: few tiles, very long reduction
                                    (300, 998, 64),        # N = 2 mod 4 (the 9998 case): 8-byte aligned k-rows, straddling chunks
                                    (998, 262, 96),        # the same on the A side (k-row A with 998 columns)
                                    (260, 388, 2030),      # K % 32 = 14: whole K tiles + register-staged tail, accumulated
                                    (188, 9998, 1440)])    # final fl::Linear of the TDS-CTC recipe (one batch row block)
@pytest.mark.parametrize("akc,bkc", [(True, False), (True, True), (False, False), (False, True)])
def test_gemm_lds_dma_path(M, N, K, akc, bkc, probe):
    """the persistent LDS-DMA kernel (K % 32 == 0, 16-byte aligned operands) against a float64 product
    and against the register-staged kernel (W2L_GEMM_GLDS=0) on the same inputs; deterministic"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = (A.double() @ Bm.double() + bias.double()).numpy()
    Ad = (A if akc else A.T.contiguous()).cuda()
    Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
    got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    assert rel(got, want) < TOL
    assert torch.equal(got, ops.gemm(Ad, Bd, akc, bkc, bias.cuda()))
    os.environ["W2L_GEMM_GLDS"] = "0"
    try:
        old = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    finally:
        os.environ.pop("W2L_GEMM_GLDS")
    assert rel(got, old.cpu().numpy()) < 1e-5
    got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda(), relu=True)
    assert rel(got, np.maximum(want, 0)) < TOL
    plain = ops.gemm(Ad, Bd, akc, bkc)      # no epilogue: the form that may split a ragged K
    assert rel(plain, (A.double() @ Bm.double()).numpy()) < TOL
    assert torch.equal(plain, ops.gemm(Ad, Bd, akc, bkc))


@pytest.mark.parametrize("M,K,N", [(300, 800, 2400),      # LDS-DMA kernel, wide epilogue
                                    (188, 1440, 9998),     # generic kernel (unaligned N), dword epilogue
                                    (37, 20, 13)])
def test_fused_epilogue_variants_equal_the_unfused_ops(M, K, N):
    """dropout folded into the Linear epilogue, the separate-addend backward-data and the out-of-place dropout are
    bit-identical to the compositions of the plain ops they replace in the TDS block"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    dy = torch.randn(M, N, generator=g).cuda()
    add = torch.randn(M, K, generator=g).cuda()
    p, seed, sid = 0.3, 12345, 7
    want = ops.linear_forward(x, w, b, relu=True)
    ops.dropout_(want, p, seed, sid)
    got = ops.linear_forward_dropout(x, w, b, True, p, seed, sid)
    assert torch.equal(got, want)
    assert 0.2 < (got == 0).float().mean().item() < 0.95  # ReLU zeros + dropped elements
    # ... and the residual join in the same epilogue (the TDS block's r2 = dropout(lin2(u)) + y1), with and without dropout
    res = torch.randn(M, N, generator=g).cuda()
    for pp in (p, 0.0):
        want2 = ops.linear_forward(x, w, b, relu=False)
        if pp > 0:
            ops.dropout_(want2, pp, seed, sid)
        assert torch.equal(ops.linear_forward_dropout_add(x, w, b, res, False, pp, seed, sid), want2 + res)
    dx_plain = ops.linear_backward(x, w, dy)[0]
    got = ops.linear_backward_data_add(dy, w, add)
    assert rel(got, (add.double() + dx_plain.double()).cpu().numpy()) < 1e-6
    src = torch.randn(4 * 1237, generator=g).cuda()
    want = src.clone()
    ops.dropout_(want, p, seed, sid)
    assert torch.equal(ops.dropout_copy(src, p, seed, sid), want)


@pytest.mark.parametrize("M,N,K", [(6016, 1440, 416), (1500, 700, 4096), (24000, 800, 2400), (800, 2400, 24000)])
def test_gemm_in_kernel_slab_reduction_is_stable(M, N, K, probe):
    """stream-K partial tiles reduced inside the GEMM launch (arrival tickets + agent-scope release/acquire):
    bit-identical to the separate fix-up launch (W2L_GEMM_INFIX=0) and to itself over many launches under load"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    Bm = (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    os.environ["W2L_GEMM_INFIX"] = "0"
    os.environ["W2L_GEMM_T160"] = "0"   # the separate fix-up launch exists for the 128x128 tile only
    try:
        ref = ops.gemm(A, Bm, True, False, bias, relu=True)
        os.environ.pop("W2L_GEMM_INFIX")
        for it in range(25):
            got = ops.gemm(A, Bm, True, False, bias, relu=True)
            assert torch.equal(got, ref), it
    finally:
        os.environ.pop("W2L_GEMM_INFIX", None)
        os.environ.pop("W2L_GEMM_T160")
    want = torch.relu(A.double() @ Bm.double() + bias.double()).cpu().numpy()
    assert rel(ref, want) < TOL


@pytest.mark.parametrize("M,N,K", [(4, 4, 32),             # one clamped tile, one K step
                                    (260, 388, 96),        # ragged edge tiles in both directions
                                    (1028, 2052, 1440),    # pure stream-K ranges crossing tiles
                                    (6016, 1440, 4320),    # TDS fc2 (c = 18): 9 x 160 columns
                                    (24000, 800, 2400),    # TDS fc2 (c = 10): 5 x 160 columns, 2 rounds + stream-K tail
                                    (800, 2400, 24000),    # dW: 5 x 160 rows (TALL), very long reduction
                                    (320, 160, 64)])       # exactly two / one 160-tiles, two K steps
@pytest.mark.parametrize("akc,bkc", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("mode", ["2", "3"])
def test_gemm_160_wide_tiles(M, N, K, akc, bkc, mode, probe):
    """the 128x160 (W2L_GEMM_T160=2 forces it) and 160x128 (=3) tile kernels on every eligible shape: float64 product,
    run-to-run bit-identity (in-kernel stream-K slab reduction), bias + ReLU, agreement with the 128x128 kernel"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11 * M + 3 * N + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = (A.double() @ Bm.double() + bias.double()).numpy()
    Ad = (A if akc else A.T.contiguous()).cuda()
    Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
    os.environ["W2L_GEMM_T160"] = mode
    try:
        got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
        for _ in range(4):
            assert torch.equal(got, ops.gemm(Ad, Bd, akc, bkc, bias.cuda()))
        gotr = ops.gemm(Ad, Bd, akc, bkc, bias.cuda(), relu=True)
        os.environ["W2L_GEMM_T160"] = "0"
        old = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    finally:
        os.environ.pop("W2L_GEMM_T160")
    assert rel(got, want) < TOL
    assert rel(gotr, np.maximum(want, 0)) < TOL
    assert rel(got, old.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(4, 4, 32), (260, 388, 96), (1028, 2052, 1440), (6016, 1440, 4320), (24000, 800, 2400),
                                    (131, 160, 64), (12000, 1120, 3360)])
@pytest.mark.parametrize("bkc", [False, True])
def test_gemm_160_a_operand_straight_into_registers(M, N, K, bkc, probe):
    """128x160 tiles with a k-contiguous A: the K loop that loads the wave's own A rows straight into the fragment registers
    (gemm160_kernel<..., ADIR>, the default) against the all-LDS loop (W2L_GEMM_ADIR=0): same k-slot assignment, same sum
    order -- bit-identical, incl. clamped edge rows, stream-K ranges and the bias + ReLU epilogue; and the float64 product"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7 * M + 5 * N + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = (A.double() @ Bm.double() + bias.double()).numpy()
    Ad = A.cuda()
    Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
    os.environ["W2L_GEMM_T160"] = "2"
    try:
        os.environ["W2L_GEMM_ADIR"] = "1"
        got = ops.gemm(Ad, Bd, True, bkc, bias.cuda())
        gotr = ops.gemm(Ad, Bd, True, bkc, bias.cuda(), relu=True)
        os.environ["W2L_GEMM_ADIR"] = "0"
        old = ops.gemm(Ad, Bd, True, bkc, bias.cuda())
        oldr = ops.gemm(Ad, Bd, True, bkc, bias.cuda(), relu=True)
    finally:
        os.environ.pop("W2L_GEMM_T160")
        os.environ.pop("W2L_GEMM_ADIR", None)
    assert rel(got, want) < TOL
    assert torch.equal(got, old) and torch.equal(gotr, oldr)


@pytest.mark.parametrize("mode", ["2", "3"])
def test_gemm_160_fused_epilogues(mode, probe):
    """dropout / mask / addend epilogues of the fl::Linear calls through the 160-wide kernels equal the 128x128 kernel's"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(160)
    M, K, N = 700, 800, 2400
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    dy = torch.randn(M, N, generator=g).cuda()
    add = torch.randn(M, K, generator=g).cuda()
    msk = torch.randn(M, K, generator=g).cuda()
    outs = {}
    for m in (mode, "0"):
        os.environ["W2L_GEMM_T160"] = m
        try:
            outs[m] = (ops.linear_forward(x, w, b, relu=True), ops.linear_forward_dropout(x, w, b, False, 0.2, 1234, 7),
                       ops.linear_backward(x, w, dy, mask_src=msk, mask_scale=1.25), ops.linear_backward_data_add(dy, w, add))
        finally:
            os.environ.pop("W2L_GEMM_T160")
    new, old = outs[mode], outs["0"]
    assert rel(new[0], old[0].cpu().numpy()) < 1e-5
    kept_new, kept_old = new[1] != 0, old[1] != 0
    assert torch.equal(kept_new, kept_old)     # same stateless mask on the same flat indices
    assert rel(new[1], old[1].cpu().numpy()) < 1e-5
    for a, c in zip(new[2], old[2]):
        assert rel(a, c.cpu().numpy()) < 1e-5
    assert rel(new[3], old[3].cpu().numpy()) < 1e-5


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches row/col swaps in the C write (guide G9)"""
    from wav2letter_amd import ops
    n = 192
    Bm = np.arange(n * n, dtype=np.float32).reshape(n, n) / 100
    got = ops.gemm(dev(np.eye(n, dtype=np.float32)), dev(Bm))
    assert (got.cpu().numpy() == Bm).all()


@pytest.mark.parametrize("M,N", [(1, 5), (63, 1), (513, 37), (777, 9998), (6016, 1440), (24000, 800), (100000, 6), (70, 2402),
                                 (957440, 27), (153600, 15), (4100, 19), (40960, 23), (8192, 63)])
def test_colsum_bias_gradient(M, N):
    """bias gradient column sums (fl::Linear / Conv2D backward): fp64 sum within 1e-4 relative, run-to-run bit-identical,
    float4 / float2 / scalar column paths and the unaligned-base case"""
    from wav2letter_amd import ops
    g = torch.Generator().manual_seed(M * 31 + N)
    x = torch.randn(M, N, generator=g) + 0.25
    want = x.double().sum(0).numpy()
    xd = dev(x)
    got = ops.colsum(xd)
    again = ops.colsum(xd)
    assert torch.equal(got, again)
    scale = np.abs(x.numpy()).sum(0) + 1.0
    assert (np.abs(got.cpu().numpy() - want) / scale).max() < 1e-5
    if N % 4 == 0 and M > 1:     # base pointer off 16-byte alignment -> narrower path, same sums
        flat = dev(torch.cat([torch.zeros(1), x.flatten()]))
        shifted = flat[1:].view(M, N)
        got2 = ops.colsum(shifted)
        assert (np.abs(got2.cpu().numpy() - want) / scale).max() < 1e-5


@pytest.mark.parametrize("M,K,N", [(37, 20, 13), (300, 800, 2400), (188, 1440, 9998)])
def test_linear_fwd_bwd(oracle, M, K, N):
    from wav2letter_amd import ops
    rng = np.random.default_rng(K)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    dy = rng.normal(size=(M, N)).astype(np.float32)
    y = ops.linear_forward(dev(x), dev(w), dev(b))
    assert rel(y, oracle.linear_fwd(x, w, b)) < TOL
    dx, dw, db = ops.linear_backward(dev(x), dev(w), dev(dy))
    odx, odw, odb = oracle.linear_bwd(x, w, dy)
    assert rel(dx, odx) < TOL and rel(dw, odw) < TOL and rel(db, odb) < TOL
    # fused ReLU forward + mask epilogue backward
    u = ops.linear_forward(dev(x), dev(w), dev(b), relu=True)
    assert rel(u, np.maximum(oracle.linear_fwd(x, w, b), 0)) < TOL


CONV_CASES = [
    # B, Cin, Cout, H, T, kw, stride, padl, padr
    (2, 1, 10, 8, 50, 21, 2, 10, 10),     # first TDS-CTC layer geometry (C2 1 10 21 1 2 1 -1 -1)
    (2, 10, 10, 8, 40, 21, 1, 10, 10),    # TDS conv c=10
    (1, 14, 14, 5, 33, 21, 1, 10, 10),    # c=14
    (2, 18, 18, 4, 30, 21, 1, 10, 10),    # c=18 (BN=32 skinny)
    (2, 10, 14, 6, 41, 21, 2, 10, 10),    # stage transition, odd T
    (2, 40, 100, 1, 60, 13, 1, 6, 6),     # conv_glu style: H=1, wide channels (128-tile)
    (1, 33, 70, 1, 45, 4, 1, 2, 2),       # even kw, odd channels (scalar loads)
    (2, 3, 5, 2, 9, 3, 1, 0, 0),          # valid (no padding)
    (1, 6, 40, 2, 64, 5, 1, 4, 0),        # asymmetric (streaming) padding
    (2, 10, 10, 80, 70, 21, 1, 10, 10),   # full TDS geometry: H = 80 (5 row blocks), 3 time blocks
    (1, 18, 18, 40, 37, 21, 1, 10, 10),   # c = 18, ragged last row block (H % 16 = 8)
    (1, 14, 18, 19, 66, 21, 2, 10, 10),   # strided stage transition, H % 16 = 3 (scalar slab loads)
    (2, 5, 7, 3, 20, 9, 1, 8, 0),         # causal padding, odd channel counts
    (2, 14, 14, 32, 100, 21, 1, 10, 10),  # persistent kernel, c = 14, 4 time blocks x 2 row blocks
    (1, 18, 18, 16, 70, 21, 1, 10, 10),   # persistent kernel, c = 18 (two MFMA column tiles, 16-frame tiles)
    (2, 10, 14, 32, 61, 21, 2, 10, 10),   # persistent kernel, strided stage transition: backward-data = 2 phase launches
    (2, 14, 18, 16, 60, 21, 2, 10, 10),   # second stage transition (18 -> 14 channel phase kernels), even T
    (1, 10, 10, 16, 47, 7, 3, 2, 3),      # stride 3: three phases with 3 / 2 / 2 taps
    (1, 12, 8, 16, 23, 4, 4, 0, 1),       # stride == kw: every phase has one tap
    (8, 10, 10, 16, 2200, 21, 1, 10, 10), # 552 tiles on 512 persistent workgroups: two tiles per workgroup + prefetch
    # more rounds than the block-Toeplitz kernels have workgroups (512 / 768): every workgroup runs >= 2 rounds, i.e. the second-slab
    # double buffering, the deferred epilogue and the addend prefetch of conv_tds_tz.hpp and the GR rounds of conv_tds_tzf.hpp --
    # what production (B >= 4, T = 1500, H = 80) always runs and the small cases above never do (round-5 advisor, medium)
    (8, 14, 14, 16, 2200, 21, 1, 10, 10),
    (8, 18, 18, 16, 1200, 21, 1, 10, 10), # two MFMA column tiles (the SSPLIT path)
    (8, 10, 14, 32, 2200, 21, 2, 10, 10), # strided 10 -> 14 forward, backward-data phase kernels (ST = 2)
    (8, 14, 18, 32, 2200, 21, 2, 10, 10), # strided 14 -> 18
    (4, 10, 10, 80, 1500, 21, 1, 10, 10), # the production geometry of the first stage at B = 4
    # conv_glu layers (H = 1, stride 1): one LDS-DMA GEMM on overlapping rows
    (2, 40, 100, 1, 60, 13, 1, 0, 0),     # valid convolution (every C4 layer but the first), K = 520 (ragged last K tile)
    (3, 33, 70, 1, 45, 4, 1, 0, 0),       # odd channel count: dword-aligned rows, N % 4 = 2 (dword epilogue)
    (2, 321, 706, 1, 64, 19, 1, 0, 0),    # C4 layer 7 channels (odd C_in), 3 utterances' worth of straddling rows dropped
    (2, 40, 400, 1, 90, 13, 1, 170, 170), # first C4 layer: 170 frames of zero padding on both sides (padded copy + remap)
    (1, 64, 64, 1, 37, 5, 1, 3, 1),       # asymmetric padding, T' not a multiple of anything
    (4, 200, 440, 1, 300, 14, 1, 0, 0),   # C4 layer 2 at reduced T: several 128-row tiles per utterance
]


@pytest.mark.parametrize("B,Cin,Cout,H,T,kw,stride,padl,padr", CONV_CASES)
def test_conv_fwd_bwd(oracle, B, Cin, Cout, H, T, kw, stride, padl, padr):
    from wav2letter_amd import ops
    rng = np.random.default_rng(Cin * 100 + Cout)
    x = rng.normal(size=(B, Cin, H, T)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, kw)) / np.sqrt(Cin * kw)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    y_ref = oracle.conv_fwd(x, w, b, stride, padl, padr)
    xd, wd = dev(to_fm(x)), dev(w_to_dev(w))
    y = ops.conv_forward(xd, wd, dev(b), stride, padl, padr)
    assert rel(from_fm(y.cpu().numpy()), y_ref) < TOL
    yr = ops.conv_forward(xd, wd, dev(b), stride, padl, padr, relu=True)
    assert rel(from_fm(yr.cpu().numpy()), np.maximum(y_ref, 0)) < TOL
    dy = rng.normal(size=y_ref.shape).astype(np.float32)
    odx, odw, odb = oracle.conv_bwd(x, w, dy, stride, padl, padr)
    dx, dw, db = ops.conv_backward(xd, wd, dev(to_fm(dy)), stride, padl, padr)
    assert rel(from_fm(dx.cpu().numpy()), odx) < TOL
    assert rel(dw.cpu().numpy(), w_to_dev(odw)) < TOL
    assert rel(db, odb) < TOL


@pytest.mark.parametrize("seed", [1, 2])
def test_gemm_and_linear_random_shapes(seed):
    """randomised shapes around the tile / schedule switches of the fp32 GEMM (128 / 160-wide tiles, stream-K tails, the dword and
    scalar epilogues of unaligned N, K not a multiple of the 32-deep K tile, single rows / columns) in all four operand layouts,
    with bias / ReLU epilogues, and fl::Linear forward + backward on the same shapes, against float64"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(4242 + seed)
    g = torch.Generator(device="cpu").manual_seed(99 + seed)
    sizes = [1, 2, 3, 5, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 159, 160, 161, 255, 256, 257, 320, 500, 513, 1000, 1440, 2047]
    for c in range(30):
        M, N, K = (int(rng.choice(sizes)) for _ in range(3))
        if c % 5 == 0:
            M = int(rng.choice([3000, 6016, 4097]))      # several rounds of tiles + a stream-K tail
        akc, bkc = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        A = torch.randn((M, K) if akc else (K, M), generator=g)
        Bm = torch.randn((N, K) if bkc else (K, N), generator=g)
        bias = torch.randn(N, generator=g) if rng.integers(0, 2) else None
        relu = bool(rng.integers(0, 2))
        want = (A.double() if akc else A.double().t()) @ (Bm.double().t() if bkc else Bm.double())
        if bias is not None:
            want = want + bias.double()
        if relu:
            want = want.clamp_min(0)
        got = ops.gemm(A.cuda(), Bm.cuda(), akc, bkc, None if bias is None else bias.cuda(), relu)
        what = f"case {c}: M={M} N={N} K={K} a_kcontig={akc} b_kcontig={bkc} bias={bias is not None} relu={relu}"
        assert rel(got, want.numpy()) < TOL, what
        # fl::Linear on the same sizes: y = x w + b, dx = dy w^T, dw = x^T dy, db = column sums
        x = torch.randn(M, K, generator=g); w = torch.randn(K, N, generator=g) / K ** 0.5; dy = torch.randn(M, N, generator=g)
        y = ops.linear_forward(x.cuda(), w.cuda(), None if bias is None else bias.cuda(), relu)
        yw = x.double() @ w.double() + (0 if bias is None else bias.double())
        assert rel(y, (yw.clamp_min(0) if relu else yw).numpy()) < TOL, what
        dx, dw, db = ops.linear_backward(x.cuda(), w.cuda(), dy.cuda())
        assert rel(dx, (dy.double() @ w.double().t()).numpy()) < TOL, what
        assert rel(dw, (x.double().t() @ dy.double()).numpy()) < TOL, what
        assert rel(db, dy.double().sum(0).numpy()) < TOL, what


@pytest.mark.parametrize("M,nin,nout", [
    (2048, 256, 320),      # 128 x 160 tiles, four of them: every tile cut by stream-K, partial column sums added by the last arriver
    (2048, 320, 256),      # 160 x 128 tiles (every wave sums its own 32 columns)
    (64, 2048, 5120),      # 512 whole tiles: column sums straight from the first tile row's registers
    (4096, 800, 2400),     # recipe widths (first TDS stage): 95 tiles over 512 workers
    (3008, 1440, 4320),    # third stage: 160 x 128 tiles
    (1504, 1120, 3364),    # N % 160 != 0: a ragged last tile column
    (1000, 300, 514),      # not eligible for the 160-wide kernel's alignment rules: the separate column-sum launch
])
def test_linear_weight_and_bias_gradient_in_one_product(M, nin, nout):
    """w2l_linear_backward_weight_bias: the bias gradient rides on the weight-gradient product (gemm160_kernel<.., CS>) -- against
    float64, and bit-identical from run to run (fixed summation order, no atomics)"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + nin + nout)
    x = torch.randn(M, nin, generator=g); w = torch.randn(nin, nout, generator=g); dy = torch.randn(M, nout, generator=g) + 0.25
    xc, wc, dyc = x.cuda(), w.cuda(), dy.cuda()
    _, dw, db = ops.linear_backward(xc, wc, dyc)
    assert rel(dw, (x.double().t() @ dy.double()).numpy()) < TOL
    assert rel(db, dy.double().sum(0).numpy()) < 1e-5
    for _ in range(3):
        _, dw2, db2 = ops.linear_backward(xc, wc, dyc)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_conv_random_geometries(oracle, seed):
    """randomised geometries around the kernel switches of the TDS convolutions (round 5: block-Toeplitz kernels for C = 10 / 14 / 18
    with H % 16 == 0, the strided layers between the stages and the one-channel first layer; everything else on the older
    generations): any kw <= 21, any left / right padding that leaves at least one output frame, T from 1 frame, ragged batches"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(1000 + seed)
    for c in range(24):
        kind = int(rng.integers(0, 5))
        H = int(rng.choice([16, 32, 48, 80, 80, 5, 8, 24, 40]))
        if kind == 0:
            Cin = Cout = int(rng.choice([10, 14, 18])); stride = 1
        elif kind == 1:
            Cin, Cout = [(10, 14), (14, 18)][int(rng.integers(0, 2))]; stride = 2
        elif kind == 2:
            Cin, Cout, stride = 1, 10, 2
        elif kind == 3:
            Cin = Cout = int(rng.choice([10, 14, 18])); stride = 1; H = 16 * int(rng.integers(1, 6))
        else:
            Cin, Cout, stride = int(rng.integers(1, 20)), int(rng.integers(1, 20)), int(rng.integers(1, 4))
        kw = int(rng.choice([21, 21, 21, 9, 1, 2, 5, 11, 20]))
        padl = int(rng.integers(0, kw))
        padr = int(rng.integers(0, kw))
        B = int(rng.integers(1, 5))
        T = int(rng.choice([1, 2, 3, 7, 23, 24, 25, 47, 48, 49, 96, 97, 130, 200]))
        if (T + padl + padr - kw) // stride + 1 < 1:
            padr = kw - 1; padl = kw - 1
        x = rng.normal(size=(B, Cin, H, T)).astype(np.float32)
        w = (rng.normal(size=(Cout, Cin, kw)) / np.sqrt(Cin * kw)).astype(np.float32)
        b = rng.normal(size=Cout).astype(np.float32)
        what = f"case {c}: B={B} Cin={Cin} Cout={Cout} H={H} T={T} kw={kw} stride={stride} padl={padl} padr={padr}"
        y_ref = oracle.conv_fwd(x, w, b, stride, padl, padr)
        xd, wd = dev(to_fm(x)), dev(w_to_dev(w))
        y = ops.conv_forward(xd, wd, dev(b), stride, padl, padr)
        assert rel(from_fm(y.cpu().numpy()), y_ref) < TOL, what
        yr = ops.conv_forward(xd, wd, dev(b), stride, padl, padr, relu=True)
        assert rel(from_fm(yr.cpu().numpy()), np.maximum(y_ref, 0)) < TOL, what
        dy = rng.normal(size=y_ref.shape).astype(np.float32)
        odx, odw, odb = oracle.conv_bwd(x, w, dy, stride, padl, padr)
        dx, dw, db = ops.conv_backward(xd, wd, dev(to_fm(dy)), stride, padl, padr)
        assert rel(from_fm(dx.cpu().numpy()), odx) < TOL, what
        assert rel(dw.cpu().numpy(), w_to_dev(odw)) < TOL, what
        assert rel(db, odb) < TOL, what


@pytest.mark.parametrize("B,Cin,Cout,H,T,kw,stride,padl,padr", [(2, 10, 14, 32, 61, 21, 2, 10, 10), (1, 14, 18, 16, 40, 21, 2, 10, 10),
                                                                (1, 10, 10, 16, 47, 7, 3, 2, 3), (1, 10, 10, 16, 40, 21, 1, 10, 10),
                                                                (2, 40, 100, 1, 60, 13, 1, 0, 0), (2, 33, 70, 1, 45, 4, 1, 2, 1),
                                                                # several rounds per workgroup: the addend prefetch (PFA) of the
                                                                # block-Toeplitz backward-data kernels and of their phase kernels
                                                                (8, 10, 10, 16, 2200, 21, 1, 10, 10), (8, 14, 14, 16, 2200, 21, 1, 10, 10),
                                                                (8, 18, 18, 16, 1200, 21, 1, 10, 10), (8, 10, 14, 32, 2200, 21, 2, 10, 10),
                                                                (8, 14, 18, 32, 2200, 21, 2, 10, 10)])
def test_conv_backward_data_accumulate_and_add(oracle, B, Cin, Cout, H, T, kw, stride, padl, padr):
    """the two fused forms of Conv2D backward-data (dx += ..., dx = add + ...) on the phase-decomposed strided path
    and the stride-1 path: equal to the plain result plus the addend"""
    import ctypes as C
    from wav2letter_amd import ops, _lib
    rng = np.random.default_rng(T)
    x = rng.normal(size=(B, Cin, H, T)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, kw)) / np.sqrt(Cin * kw)).astype(np.float32)
    To = (T + padl + padr - kw) // stride + 1
    dy = rng.normal(size=(B, Cout, H, To)).astype(np.float32)
    odx, _, _ = oracle.conv_bwd(x, w, dy, stride, padl, padr)
    xd, wd, dyd = dev(to_fm(x)), dev(w_to_dev(w)), dev(to_fm(dy))
    d = ops.conv_desc(xd, wd, stride, padl, padr)
    L = _lib.lib()
    base = rng.normal(size=x.shape).astype(np.float32)
    acc = dev(to_fm(base))
    ops.check(L.w2l_conv_backward_data(C.byref(d), ops._p(dyd), ops._p(wd), ops._p(acc), 1, ops._s()), "bwd data accumulate")
    assert rel(from_fm(acc.cpu().numpy()), odx + base) < TOL
    add = dev(to_fm(base))
    out = torch.empty_like(add)
    ops.check(L.w2l_conv_backward_data_add(C.byref(d), ops._p(dyd), ops._p(wd), ops._p(add), ops._p(out), ops._s()), "bwd data add")
    assert rel(from_fm(out.cpu().numpy()), odx + base) < TOL
    ops.check(L.w2l_conv_backward_data_add(C.byref(d), ops._p(dyd), ops._p(wd), ops._p(add), ops._p(add), ops._s()), "bwd data add in place")
    assert rel(from_fm(add.cpu().numpy()), odx + base) < TOL


TDS_RS3_SHAPES = [(10, 48, 2, 80), (18, 12, 2, 80), (10, 50, 2, 80), (18, 15, 3, 80), (10, 1, 1, 80), (18, 2, 2, 80), (10, 129, 3, 80),
                  (18, 188, 5, 80), (10, 750, 3, 80), (10, 64, 1, 80), (18, 46, 7, 80), (10, 331, 9, 80), (18, 97, 33, 80), (10, 77, 5, 8),
                  (18, 150, 2, 24), (10, 2100, 2, 16), (14, 24, 2, 80), (14, 77, 2, 80), (14, 375, 3, 80), (14, 1, 1, 80), (14, 33, 5, 16), (14, 200, 9, 8)]


def _tds_conv_ref64(x, w, b, dy, add, kw, padl):
    import torch.nn.functional as F
    xr = x.double().permute(0, 3, 2, 1).requires_grad_(True)          # [B][C][H][T]
    wr = w.double().permute(2, 1, 0)[:, :, None, :]
    yr = F.conv2d(F.pad(xr, (padl, kw - 1 - padl)), wr, b.double())
    yr.backward(dy.double().permute(0, 3, 2, 1))
    return yr.permute(0, 3, 2, 1).detach(), xr.grad.permute(0, 3, 2, 1) + add.double()


@pytest.mark.parametrize("Cc,T,B,H", TDS_RS3_SHAPES)
def test_tds_conv_streamed_wave_specialised_kernel(Cc, T, B, H):
    """conv_tds_rs3.hpp (the TDS convolution proper, C = 10 / 14 / 18, H % 8 == 0): forward with bias, with and without the fused
    ReLU, and backward-data with the fused addend, every element against a float64 convolution; segment cuts in the
    middle of utterances, utterances shorter than one tile, one- and two-frame inputs, short kernels, causal padding;
    run-to-run identical (the overlap-add order is program order)"""
    import ctypes as C
    from wav2letter_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for kw, padl in ((21, 10), (21, 20), (9, 0)):
        if kw - 1 - padl >= T + padl and T < 3:
            continue
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, padl, kw - 1 - padl)
        g = torch.Generator(device="cpu").manual_seed(Cc * 1000 + T + kw)
        x = torch.randn(B, T, H, Cc, generator=g).cuda()
        w = (torch.randn(kw, Cc, Cc, generator=g) / (kw * Cc) ** 0.5).cuda()
        b = torch.randn(Cc, generator=g).cuda()
        dy = torch.randn(B, T, H, Cc, generator=g).cuda()
        add = torch.randn(B, T, H, Cc, generator=g).cuda()
        ry, rdx = _tds_conv_ref64(x, w, b, dy, add, kw, padl)
        outs = []
        for rep in range(2):
            y = torch.full_like(x, float("nan")); yr = torch.full_like(x, float("nan")); dx = torch.full_like(x, float("nan"))
            assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, s) == 0
            assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), yr.data_ptr(), 1, s) == 0
            assert L.w2l_conv_backward_data_add(C.byref(d), dy.data_ptr(), w.data_ptr(), add.data_ptr(), dx.data_ptr(), s) == 0
            torch.cuda.synchronize()
            outs.append((y, yr, dx))
        y, yr, dx = outs[0]
        assert rel(y, ry.cpu().numpy()) < TOL
        assert rel(yr, torch.relu(ry).cpu().numpy()) < TOL
        assert rel(dx, rdx.cpu().numpy()) < TOL
        assert all(torch.equal(a, b2) for a, b2 in zip(outs[0], outs[1]))


@pytest.mark.parametrize("Cc,T,B", [(10, 331, 3), (18, 97, 5), (14, 77, 2)])
def test_tds_conv_kernel_generations_agree(Cc, T, B, probe):
    """the previous generations stay selectable in the probe library (W2L_TDS_RS3_OFF: the cooperative role-swapped kernel,
    W2L_TDS_RS_OFF: conv_tds.hip's 16-wide tiles) and agree with the product kernels"""
    import ctypes as C
    import os
    from wav2letter_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    H, kw = 80, 21
    d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
    g = torch.Generator(device="cpu").manual_seed(Cc + T)
    x = torch.randn(B, T, H, Cc, generator=g).cuda()
    w = (torch.randn(kw, Cc, Cc, generator=g) / (kw * Cc) ** 0.5).cuda()
    b = torch.randn(Cc, generator=g).cuda()

    def fwd(Lx):
        y = torch.empty_like(x)
        assert Lx.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
        torch.cuda.synchronize()
        return y
    want = fwd(_lib.lib())
    for sw in ("W2L_TDS_RS3_OFF", "W2L_TDS_RS_OFF"):
        os.environ[sw] = "1"
        try:
            got = fwd(probe)
        finally:
            os.environ.pop(sw)
        assert rel(got, want.cpu().numpy()) < TOL


@pytest.mark.parametrize("B,Cin,Cout,T,kw,padl,padr", [(2, 40, 100, 60, 13, 0, 0), (2, 321, 706, 64, 19, 0, 0), (2, 40, 400, 90, 13, 170, 170)])
def test_conv_glu_overlapping_rows_equals_implicit_gemm(B, Cin, Cout, T, kw, padl, padr, probe):
    """the overlapping-row LDS-DMA convolution against the register-staged implicit-GEMM kernels (W2L_CONV_GLDS=0) on
    the same inputs: forward (+bias, ReLU), backward-data, backward-filter, bias gradient; run-to-run identical"""
    import os
    from wav2letter_amd import ops
    g = torch.Generator().manual_seed(Cin + T)
    x = torch.randn(B, T, 1, Cin, generator=g).cuda()
    w = (torch.randn(kw, Cin, Cout, generator=g) / (kw * Cin) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    To = T + padl + padr - kw + 1
    dy = torch.randn(B, To, 1, Cout, generator=g).cuda()
    outs = {}
    for mode in ("1", "0"):
        os.environ["W2L_CONV_GLDS"] = mode
        try:
            y = ops.conv_forward(x, w, b, 1, padl, padr, relu=True)
            dx, dw, db = ops.conv_backward(x, w, dy, 1, padl, padr)
            outs[mode] = (y, dx, dw, db)
            if mode == "1":
                again = ops.conv_backward(x, w, dy, 1, padl, padr)
                assert torch.equal(again[0], dx) and torch.equal(again[1], dw)
        finally:
            os.environ.pop("W2L_CONV_GLDS")
    for a, c in zip(outs["1"], outs["0"]):
        assert a.shape == c.shape
        assert rel(a, c.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,T,kw,padl,padr", [(64, 388, 852, 2000, 21, 0, 0),      # C4 layer 9 at full size
                                                        (64, 40, 400, 2000, 13, 170, 170),  # C4 layer 1 (zero padding)
                                                        (16, 826, 1816, 2000, 29, 0, 0)])   # C4 layer 17 (batch reduced)
def test_conv_glu_full_size_adjoint_identities(B, Cin, Cout, T, kw, padl, padr):
    """BASELINE config 4 layer sizes, where the oracle cannot run: the convolution is bilinear in (x, w), so
    <conv(x, w), dy> = <x, backward_data(dy, w)> = <w, backward_filter(x, dy)> -- one number computed three ways through
    three different GEMMs of the overlapping-row path -- and the bias gradient is the column sum of dy"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cuda").manual_seed(Cin + kw)
    x = torch.randn(B, T, 1, Cin, generator=g, device="cuda")
    w = torch.randn(kw, Cin, Cout, generator=g, device="cuda") / (kw * Cin) ** 0.5
    To = T + padl + padr - kw + 1
    dy = torch.randn(B, To, 1, Cout, generator=g, device="cuda")
    y = ops.conv_forward(x, w, None, 1, padl, padr)
    dx, dw, db = ops.conv_backward(x, w, dy, 1, padl, padr)
    a = (y.double() * dy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    scale = (y.double().abs() * dy.double().abs()).sum().item()
    assert abs(a - b) < 1e-5 * scale and abs(a - c) < 1e-5 * scale, (a, b, c, scale)
    assert rel(db, dy.double().sum(dim=(0, 1, 2)).cpu().numpy()) < 1e-5
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()


def test_golden_conv1d_on_device():
    """the reference's own golden vector (Conv1dTest.cpp:30-104) through the HIP conv"""
    import json, os
    from wav2letter_amd import ops
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "conv1d_golden.json")))
    T, H, Cc, kw = g["T"], g["groups"], g["channels"] // g["groups"], g["kernelSize"]
    x = np.array(g["input"], np.float32).reshape(1, T, H, Cc)          # streaming layout == frame-major
    w = np.array(g["weights"], np.float32).reshape(Cc, kw, Cc).transpose(1, 2, 0)  # [co][k][ci] -> [k][ci][co]
    y = ops.conv_forward(dev(x), dev(np.ascontiguousarray(w)), dev(np.array(g["bias"], np.float32)), 1,
                         g["leftPadding"], g["rightPadding"])
    assert np.abs(y.cpu().numpy().reshape(-1) - np.array(g["target"])).max() < 2e-3


@pytest.mark.parametrize("B,inner", [(3, 4 * 1237), (37, 1440), (5, 8192), (2, 4 * 40001),
                                     # every chunks-per-thread instance of the one-pass kernels (round 5): 1 ... 8 x 1024 floats, ragged last chunk
                                     (7, 4), (6, 1028), (5, 2800), (4, 4000), (3, 5600), (3, 6800), (2, 8188)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_residual_dropout_layernorm(oracle, p, B, inner):
    """utterance-sized groups (two-level deterministic reduction) and frame-sized groups (one-pass kernel)"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(5)
    a = np.maximum(rng.normal(size=(B, inner)), 0).astype(np.float32)
    x = rng.normal(size=(B, inner)).astype(np.float32)
    gb = np.array([1.3, -0.2], np.float32)
    ad = dev(a)
    y, r, mr = ops.residual_layernorm_forward(ad, dev(x), dev(gb), B, 1e-5, p, 77, 5)
    a_drop = oracle.dropout(a.reshape(-1), p, 77, 5).reshape(a.shape) if p > 0 else a
    assert (ad.cpu().numpy() == a_drop).all()          # identical mask (integer hash), bit-exact
    r_ref = a_drop + x
    assert rel(r, r_ref) < 1e-6
    y_ref = oracle.layernorm_fwd(r_ref, B, 1.3, -0.2, 1e-5).reshape(B, inner)
    assert rel(y, y_ref) < TOL
    dy = rng.normal(size=(B, inner)).astype(np.float32)
    odr, odg, odb = oracle.layernorm_bwd(r_ref, dy, B, 1.3, 1e-5)
    sc = 1.0 / (1.0 - p)
    dr, dgb, dmask = ops.layernorm_backward(r, dev(dy), dev(gb), mr, B, mask_src=ad, mask_scale=sc)
    assert rel(dr, odr.reshape(B, inner)) < TOL
    assert abs(dgb[0].item() - odg) < TOL * max(1, abs(odg)) * 10
    assert abs(dgb[1].item() - odb) < TOL * max(1, abs(odb)) * 10
    assert rel(dmask, odr.reshape(B, inner) * (a_drop > 0) * sc) < TOL


def test_tds_block_composed_from_kernels(oracle):
    """forward + backward of one TDS block built from the C-ABI ops == oracle composition"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(6)
    B, Cc, H, T, kw, l2 = 2, 10, 6, 24, 21, 96
    p = refnet.TDSParams(Cc, kw, H, l2=l2, rng=rng)
    x = rng.normal(size=(B, Cc, H, T)).astype(np.float32)
    dout = rng.normal(size=x.shape).astype(np.float32)
    out_ref, saved = refnet.tds_fwd(x, p, 10, 10, "all", keep=True)
    dx_ref, g_ref = refnet.tds_bwd(dout, p, saved, 10, 10, "all")

    xd = dev(to_fm(x))
    wc = dev(w_to_dev(p.wc))
    gb1 = dev(np.array([p.g1, p.b1n], np.float32)); gb2 = dev(np.array([p.g2, p.b2n], np.float32))
    w1, b1, w2, b2 = dev(p.w1), dev(p.b1), dev(p.w2), dev(p.b2)
    # forward
    a = ops.conv_forward(xd, wc, dev(p.bc), 1, 10, 10, relu=True)
    y, r, mr1 = ops.residual_layernorm_forward(a, xd, gb1, B)
    M, l = B * T, H * Cc
    u = ops.linear_forward(y.view(M, l), w1, b1, relu=True)
    v = ops.linear_forward(u, w2, b2)
    out, s, mr2 = ops.residual_layernorm_forward(v.view(B, T, H, Cc), y, gb2, B)
    assert rel(from_fm(out.cpu().numpy()), out_ref) < TOL
    # backward
    dd = dev(to_fm(dout))
    ds, dgb2, _ = ops.layernorm_backward(s, dd, gb2, mr2, B)
    du, dw2, db2 = ops.linear_backward(u, w2, ds.view(M, l), mask_src=u, mask_scale=1.0)
    dz, dw1, db1 = ops.linear_backward(y.view(M, l), w1, du)
    dy = ds + dz.view(B, T, H, Cc)
    dr, dgb1, da = ops.layernorm_backward(r, dy, gb1, mr1, B, mask_src=a, mask_scale=1.0)
    dxc, dwc, dbc = ops.conv_backward(xd, wc, da, 1, 10, 10)
    dx = dr + dxc
    assert rel(from_fm(dx.cpu().numpy()), dx_ref) < TOL
    assert rel(dwc.cpu().numpy(), w_to_dev(g_ref["wc"])) < TOL
    assert rel(dbc, g_ref["bc"]) < TOL
    assert rel(dw1, g_ref["w1"]) < TOL and rel(db1, g_ref["b1"]) < TOL
    assert rel(dw2, g_ref["w2"]) < TOL and rel(db2, g_ref["b2"]) < TOL
    assert abs(dgb1[0].item() - g_ref["g1"]) < 1e-3 * max(1, abs(g_ref["g1"]))
    assert abs(dgb2[1].item() - g_ref["b2n"]) < 1e-3 * max(1, abs(g_ref["b2n"]))


@pytest.mark.parametrize("K,N", [(15, 4), (520, 400), (2800, 442), (333, 77), (6099, 706)])
def test_weightnorm_columns(oracle, K, N):
    """fl::WeightNorm over the internal [K][Nout] layout (WN 3 C: K = kw*C_in, WN 0 L: K = in): w, dv, dg against the
    oracle; two-pass column reductions are run-to-run identical"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(K + N)
    v = rng.normal(size=(K, N)).astype(np.float32)
    g = rng.uniform(0.5, 2.0, size=N).astype(np.float32)
    dw = rng.normal(size=(K, N)).astype(np.float32)
    w, norm = ops.weightnorm_forward(dev(v), dev(g))
    assert rel(w, oracle.weightnorm_fwd(v, g, K, N, 1)) < TOL
    assert rel(norm, np.sqrt((v.astype(np.float64) ** 2).sum(0))) < 1e-5
    dv, dg = ops.weightnorm_backward(dev(v), dev(g), norm, dev(dw))
    odv, odg = oracle.weightnorm_bwd(v, g, dw, K, N, 1)
    assert rel(dv, odv) < TOL and rel(dg, odg) < TOL
    dv2, dg2 = ops.weightnorm_backward(dev(v), dev(g), norm, dev(dw))
    assert torch.equal(dv, dv2) and torch.equal(dg, dg2)


def test_glu_transpose_sgd(oracle):
    from wav2letter_amd import ops
    rng = np.random.default_rng(8)
    M, half = 37, 13
    x = rng.normal(size=(M, 2 * half)).astype(np.float32)
    dy = rng.normal(size=(M, half)).astype(np.float32)
    assert rel(ops.glu_forward(dev(x)), oracle.glu_fwd(x, M, half, 1).reshape(M, half)) < TOL
    assert rel(ops.glu_backward(dev(x), dev(dy)), oracle.glu_bwd(x, dy, M, half, 1)) < TOL
    t = rng.normal(size=(3, 45, 70)).astype(np.float32)
    assert (ops.transpose(dev(t)).cpu().numpy() == t.transpose(0, 2, 1)).all()
    n = 10007
    p0 = rng.normal(size=n).astype(np.float32); g = rng.normal(size=n).astype(np.float32)
    v0 = rng.normal(size=n).astype(np.float32)
    pd, vd = dev(p0), dev(v0)
    ops.sgd_step_(pd, dev(g), vd, 0.3, 0.5, grad_scale=0.25, max_grad_norm=1.0)
    gs = g.astype(np.float64) * 0.25
    coef = min(1.0, 1.0 / (np.linalg.norm(gs) + 1e-6))
    v1 = 0.5 * v0 + gs * coef
    assert rel(vd, v1) < 1e-5 and rel(pd, p0 - 0.3 * v1) < 1e-5


def test_gemm_throughput_report():
    """not pass/fail: prints achieved TFLOP/s on the TDS-CTC fc shapes (fp32 MFMA peak 157.3)"""
    from wav2letter_amd import ops
    torch.manual_seed(0)
    for (M, K, N, tag) in [(24000, 800, 2400, "fc1 s1"), (24000, 2400, 800, "fc2 s1"), (12000, 1120, 3360, "fc1 s2"),
                           (6016, 1440, 4320, "fc1 s3"), (6016, 1440, 9998, "final"), (4096, 4096, 4096, "4096^3")]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda"); dy = torch.randn(M, N, device="cuda")
        for name, fn in [("fwd", lambda: ops.linear_forward(x, w, b, relu=True)),
                         ("bwd", lambda: ops.linear_backward(x, w, dy))]:
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            fl = 2.0 * M * K * N * (1 if name == "fwd" else 2)
            print(f"\n[gemm] {tag:8s} {name} M={M} K={K} N={N}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s", end="")
    print()


@pytest.mark.parametrize("B,ns,F,power", [(3, 19200, 80, False), (2, 24000, 40, False), (1, 16000 + 37, 80, True), (4, 400, 80, False)])
def test_mfsc_features_on_device(oracle, B, ns, F, power):
    """log-mel front end (one overlapping-row GEMM for the spectra of all frames, |.|, mel GEMM, log + transposition to
    [B][NFEAT][T]) against the frame-by-frame fp64 restatement; 1e-4 of the largest feature magnitude; ragged sample
    counts (not a whole number of strides) and the one-frame utterance"""
    from wav2letter_amd.features import Mfsc
    rng = np.random.default_rng(ns + F)
    audio = (rng.normal(size=(B, ns)) * 3000.0).astype(np.float32)
    fe = Mfsc(num_filters=F, use_power=power)
    got = fe(dev(audio))
    T = fe.num_frames(ns)
    assert got.shape == (B, F, T)
    want = np.stack([oracle.mfsc(audio[b], F, use_power=power).T for b in range(B)])
    assert np.abs(got.cpu().numpy() - want).max() < 1e-4 * np.abs(want).max()
    assert torch.equal(got, fe(dev(audio)))


def test_product_library_ignores_kernel_variant_switches():
    """round-1 verdict / advisor: an inherited W2L_* variable must not change (let alone corrupt) training.  The shipped
    libw2l_hip.so never reads the environment: with the garbage-producing ablation switches set, results stay exact;
    the probe build (libw2l_hip_probe.so) is the only one that honours them"""
    import os
    import subprocess
    import sys
    code = (
        "import os, torch, numpy as np\n"
        "from wav2letter_amd import ops\n"
        "g = torch.Generator(device='cpu').manual_seed(1)\n"
        "A = torch.randn(512, 256, generator=g); B = torch.randn(256, 384, generator=g)\n"
        "got = ops.gemm(A.cuda(), B.cuda(), True, False).cpu().double()\n"
        "want = A.double() @ B.double()\n"
        "print('REL', float((got - want).abs().max() / want.abs().max()))\n")
    env = dict(os.environ, W2L_GEMM_ABL="1", W2L_GEMM_ABLBUF="1", W2L_FCC_ABL="1", W2L_TDS_ABL="7", W2L_GEMM_GLDS="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rel_ = float([l for l in out.stdout.splitlines() if l.startswith("REL")][0].split()[1])
    assert rel_ < TOL
    # and the product library does not even contain the ablation / experimental kernels
    so = os.path.join(root, "wav2letter_amd", "libw2l_hip.so")
    blob = open(so, "rb").read()
    assert b"W2L_GEMM_ABL" not in blob and b"W2L_FCC_ABL" not in blob


# ----------------------------------------------------------------------------------------------
# mixed precision (BASELINE config 3): bf16 multiply / fp32 accumulate GEMM, fp32 storage
# ----------------------------------------------------------------------------------------------
BF16_TOL = 1e-2   # stated tolerance: operands rounded to bf16 (8 significant bits, relative 2^-9 per element), fp32
                  # accumulation: |error| / max|reference| stays below 1e-2 for reductions up to ~1e4 terms


@pytest.fixture
def bf16_matmul():
    from wav2letter_amd import _lib
    prev = _lib.lib().w2l_set_matmul_precision(1)
    try:
        yield
    finally:
        _lib.lib().w2l_set_matmul_precision(prev)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (130, 250, 70), (257, 129, 33), (1200, 1520, 96), (11968, 1200, 1200),
                                    (1200, 1200, 11968), (188, 9998, 2160)])
@pytest.mark.parametrize("akc,bkc", [(True, False), (True, True), (False, False), (False, True)])
def test_gemm_bf16_all_layouts(bf16_matmul, M, N, K, akc, bkc):
    """every operand layout of the fl::Linear calls (forward k-contiguous x k-rows, dX both k-contiguous, dW both k-rows
    with the split-K slabs) against a float64 product AND against a float64 product of bf16-rounded operands (the
    kernel's only approximation is that rounding): 1e-2 / 1e-5; bias + ReLU epilogue; run-to-run identical"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = (A.double() @ Bm.double() + bias.double()).numpy()
    want_r = (A.bfloat16().double() @ Bm.bfloat16().double() + bias.double()).numpy()
    Ad = (A if akc else A.T.contiguous()).cuda()
    Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
    got = ops.gemm(Ad, Bd, akc, bkc, bias.cuda())
    assert rel(got, want) < BF16_TOL
    assert rel(got, want_r) < 2e-5          # exact up to fp32 accumulation order
    assert torch.equal(got, ops.gemm(Ad, Bd, akc, bkc, bias.cuda()))
    assert rel(ops.gemm(Ad, Bd, akc, bkc, bias.cuda(), relu=True), np.maximum(want, 0)) < BF16_TOL
    plain = ops.gemm(Ad, Bd, akc, bkc)       # no epilogue: the form that may split K over slabs
    assert rel(plain, (A.bfloat16().double() @ Bm.bfloat16().double()).numpy()) < 2e-5
    assert torch.equal(plain, ops.gemm(Ad, Bd, akc, bkc))


@pytest.mark.parametrize("seed", [1, 2])
def test_gemm_bf16_random_shapes(bf16_matmul, seed):
    """the bf16-operand GEMM behind w2l_set_matmul_precision over random shapes and layouts (odd K, single rows, N % 8 != 0,
    several rounds of tiles): exact up to fp32 accumulation order against a float64 product of the bf16-rounded operands"""
    from wav2letter_amd import ops
    rng = np.random.default_rng(777 + seed)
    g = torch.Generator(device="cpu").manual_seed(55 + seed)
    sizes = [1, 2, 3, 7, 31, 32, 33, 64, 65, 96, 127, 128, 129, 250, 256, 257, 500, 1000, 1200, 1521]
    for c in range(24):
        M, N, K = (int(rng.choice(sizes)) for _ in range(3))
        if c % 6 == 0:
            M = int(rng.choice([3008, 6016, 11968]))
        akc, bkc = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        A = torch.randn(M, K, generator=g)
        Bm = torch.randn(K, N, generator=g) / K ** 0.5
        bias = torch.randn(N, generator=g) if rng.integers(0, 2) else None
        relu = bool(rng.integers(0, 2))
        want = A.bfloat16().double() @ Bm.bfloat16().double()
        exact = A.double() @ Bm.double()
        if bias is not None:
            want, exact = want + bias.double(), exact + bias.double()
        if relu:
            want, exact = want.clamp_min(0), exact.clamp_min(0)
        Ad = (A if akc else A.T.contiguous()).cuda()
        Bd = (Bm.T.contiguous() if bkc else Bm).cuda()
        got = ops.gemm(Ad, Bd, akc, bkc, None if bias is None else bias.cuda(), relu)
        what = f"case {c}: M={M} N={N} K={K} a_kcontig={akc} b_kcontig={bkc} bias={bias is not None} relu={relu}"
        # (shapes the bf16 kernels do not take -- K below a tile, unaligned operands -- run on the fp32 GEMM: the unrounded product)
        assert rel(got, want.numpy()) < 2e-5 or rel(got, exact.numpy()) < TOL, what
        assert torch.equal(got, ops.gemm(Ad, Bd, akc, bkc, None if bias is None else bias.cuda(), relu)), what


def test_linear_ops_bf16_epilogues(oracle, bf16_matmul):
    """the fl::Linear entry points of the TDS block under bf16 multiplies: dropout epilogue mask identical to the fp32
    path (same stateless hash), mask / addend epilogues, weight gradient; against the fp64 oracle at the bf16 tolerance"""
    from wav2letter_amd import _lib, ops
    rng = np.random.default_rng(5)
    M, K, N = 300, 96, 160
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(K, N)) / K ** 0.5).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    y = ops.linear_forward(dev(x), dev(w), dev(b), relu=True)
    assert rel(y, np.maximum(oracle.linear_fwd(x, w, b), 0)) < BF16_TOL
    yd = ops.linear_forward_dropout(dev(x), dev(w), dev(b), True, 0.3, 7, 3)
    prev = _lib.lib().w2l_set_matmul_precision(0)
    yd32 = ops.linear_forward_dropout(dev(x), dev(w), dev(b), True, 0.3, 7, 3)
    _lib.lib().w2l_set_matmul_precision(prev)
    assert torch.equal(yd == 0, yd32 == 0) or ((yd == 0) != (yd32 == 0)).float().mean().item() < 1e-3   # ReLU zeros may differ at the rounding level
    assert rel(yd, yd32.cpu().numpy()) < BF16_TOL
    dy = rng.normal(size=(M, N)).astype(np.float32)
    dx, dw, db = ops.linear_backward(dev(x), dev(w), dev(dy))
    odx, odw, odb = oracle.linear_bwd(x, w, dy)
    assert rel(dx, odx) < BF16_TOL and rel(dw, odw) < BF16_TOL and rel(db, odb) < 1e-4   # the bias gradient is an fp32 column sum
    add = rng.normal(size=(M, K)).astype(np.float32)
    assert rel(ops.linear_backward_data_add(dev(dy), dev(w), dev(add)), odx + add) < BF16_TOL


# ----------------------------------------------------------------------------------------------
# mixed precision with bf16 OPERAND STORAGE (round 3): convert.hip + gemm_bf16g.hpp
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1, 1), (5, 3), (64, 64), (70, 130), (187, 2160), (300, 9998), (1200, 77)])
def test_bf16_convert_images(rows, cols):
    """both images of w2l_bf16_convert against torch's fp32 -> bfloat16 conversion (round to nearest even) BIT FOR BIT, zero
    padding of either leading dimension to the next multiple of 64 included; each image alone gives the same bytes"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows * 31 + cols)
    x = (torch.randn(rows, cols, generator=g) * 10.0 ** torch.randint(-3, 4, (rows, cols), generator=g)).cuda()
    want = x.to(torch.bfloat16)
    rm, tr = ops.bf16_convert(x, True, True)
    colsP, rowsP = (cols + 63) // 64 * 64, (rows + 63) // 64 * 64
    assert rm.shape == (rows, colsP) and tr.shape == (cols, rowsP)
    assert torch.equal(rm[:, :cols].view(torch.int16), want.view(torch.int16))
    assert torch.equal(tr[:, :rows].view(torch.int16), want.T.contiguous().view(torch.int16))
    assert (rm[:, cols:].view(torch.int16) == 0).all() and (tr[:, rows:].view(torch.int16) == 0).all()
    rm2, none = ops.bf16_convert(x, True, False)
    none2, tr2 = ops.bf16_convert(x, False, True)
    assert none is None and none2 is None
    assert torch.equal(rm2.view(torch.int16), rm.view(torch.int16)) and torch.equal(tr2.view(torch.int16), tr.view(torch.int16))


def test_bf16_convert_multi():
    """w2l_bf16_convert_multi: several matrices in one launch, each image bit-identical to its own w2l_bf16_convert"""
    import ctypes as C
    from wav2letter_amd import _lib, ops
    g = torch.Generator().manual_seed(4)
    shapes = [(1024, 1024), (64, 4096), (100, 37), (1, 64), (300, 129), (65, 65), (7, 1000), (129, 2)]
    xs = [torch.randn(r, c, generator=g).cuda() for r, c in shapes]
    want = [ops.bf16_convert(x, True, True) for x in xs]
    for n in (1, 3, 8):
        rms, trs, arr = [], [], (_lib.Bf16ConvertDesc * n)()
        for i in range(n):
            r, c = shapes[i]
            rm = torch.full((r, (c + 63) // 64 * 64), -1, dtype=torch.bfloat16, device="cuda")
            tr = torch.full((c, (r + 63) // 64 * 64), -1, dtype=torch.bfloat16, device="cuda") if i != 1 else None
            rms.append(rm); trs.append(tr)
            arr[i] = _lib.Bf16ConvertDesc(x=xs[i].data_ptr(), rows=r, cols=c, ldx=c, rowMajor=rm.data_ptr(), ldRows=rm.shape[1],
                                          transposed=tr.data_ptr() if tr is not None else None, ldTrans=tr.shape[1] if tr is not None else 0)
        assert _lib.lib().w2l_bf16_convert_multi(n, arr, torch.cuda.current_stream().cuda_stream) == 0
        for i in range(n):
            assert torch.equal(rms[i].view(torch.int16), want[i][0].view(torch.int16))
            if trs[i] is not None:
                assert torch.equal(trs[i].view(torch.int16), want[i][1].view(torch.int16))
    assert _lib.lib().w2l_bf16_convert_multi(9, arr, torch.cuda.current_stream().cuda_stream) != 0


def _bf16_ref(a):
    return a.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(4, 4, 8), (128, 128, 64), (260, 388, 96), (1028, 2052, 1440), (187 * 4, 1200, 1200),
                                    (11968, 2160, 2160), (3008, 9998, 1024), (640, 136, 24000), (2160, 9998, 748),
                                    (1200, 1200, 11968), (2100, 2100, 6100), (4000, 3100, 200), (2160, 2160, 11968),
                                    (11968, 1200, 1200), (513, 6200, 4000)])
def test_gemm_bf16_operand_storage(M, N, K):
    """C = A B^T on bf16 images against the float64 product of the SAME bf16-rounded operands (the kernel's only rounding is
    the fp32 accumulation: 2e-5 of the largest magnitude), run-to-run determinism (stream-K slabs are added in range order),
    and the shapes of the three products of config 3 / config 5 layers incl. a ragged K and an N that is not a multiple of 4.
    The larger shapes run on the 256 x 256 tile kernel: whole tiles only (K = 200), stream-K ranges only (2100 x 2100 x 6100,
    the weight-gradient shape), a full round plus ranges (11968 x 2160), the dword epilogue (N = 9998), one row of tiles"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Ab, _ = ops.bf16_convert(A)
    Bb, _ = ops.bf16_convert(B)
    want = (_bf16_ref(A) @ _bf16_ref(B).T).cpu().numpy()
    got = ops.gemm_bf16(Ab, Bb, K)
    assert rel(got, want) < 2e-5
    assert torch.equal(got, ops.gemm_bf16(Ab, Bb, K))
    # against the UNROUNDED product: the stated bf16 tolerance
    assert rel(got, (A.double() @ B.double().T).cpu().numpy()) < BF16_TOL
    wb = want + bias.double().cpu().numpy()
    assert rel(ops.gemm_bf16(Ab, Bb, K, bias=bias, relu=True), np.maximum(wb, 0)) < 2e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 100), (1200, 1200, 11968), (1520, 1200, 3000), (300, 2160, 2160), (11968, 1200, 1200),
                                   (64, 9998, 2160), (1024, 1024, 3008), (40, 24, 8)])
def test_gemm_bf16_k_major_operands(M, N, K):
    """round 4: the bf16 GEMM reads k-MAJOR operands in place (A stored [K][M], B stored [K][N]: the row-major image of an
    activation as the operand of a weight gradient x^T dy, the row-major image of a weight as the B operand of x w) through the
    LDS transpose read -- no transposed bf16 image.  All four operand combinations against the k-contiguous kernel on the
    transposed copies of the SAME bf16 values: the MFMA sequence is the same, so the results must be BIT-IDENTICAL; and against
    the float64 product of the rounded operands at 2e-5."""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Ab, _ = ops.bf16_convert(A)                     # [M][Kp], zero padded
    Bb, _ = ops.bf16_convert(B)
    M8, N8 = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    At = torch.zeros(K, M8, dtype=torch.bfloat16, device="cuda")      # k-major: [K][M], row stride a multiple of 8
    At[:, :M] = Ab[:, :K].T
    Bt = torch.zeros(K, N8, dtype=torch.bfloat16, device="cuda")
    Bt[:, :N] = Bb[:, :K].T
    base = ops.gemm_bf16(Ab, Bb, K)
    want = (_bf16_ref(A) @ _bf16_ref(B).T).cpu().numpy()
    assert rel(base, want) < 2e-5
    for ta, tb in ((True, True), (True, False), (False, True)):
        got = ops.gemm_bf16_ex(At if ta else Ab, Bt if tb else Bb, M, N, K, a_kmajor=ta, b_kmajor=tb)
        assert rel(got, want) < 2e-5, (ta, tb)
        assert torch.equal(got, base), (ta, tb, float((got - base).abs().max()))
    got = ops.gemm_bf16_ex(At, Bt, M, N, K, a_kmajor=True, b_kmajor=True, bias=bias, relu=True)
    assert rel(got, np.maximum(want + bias.double().cpu().numpy(), 0)) < 2e-5


@pytest.mark.parametrize("B,T,Cin,Cout,kw,padl,padr", [(2, 50, 80, 64, 3, 1, 1), (3, 37, 32, 48, 5, 2, 2), (2, 40, 512, 1024, 3, 1, 1),
                                                       (2, 33, 64, 96, 7, 0, 0), (1, 90, 48, 32, 13, 12, 0)])
def test_conv_overlapping_rows_bf16_mode(B, T, Cin, Cout, kw, padl, padr):
    """the kw x 1 convolutions (WN-Conv front end of the Transformer recipe, conv_glu layers) under w2l_set_matmul_precision(1):
    forward and backward-data run as overlapping-row GEMMs on bf16 images (conv.hip); against the float64 convolution of the
    SAME bf16-rounded operands (2e-5 of the largest magnitude: fp32 accumulation only) and the stated 1e-2 against the unrounded
    one; the filter gradient stays on the fp32 path (1e-4).  The mode is restored afterwards."""
    import torch.nn.functional as F
    from wav2letter_amd import _lib, ops
    g = torch.Generator().manual_seed(Cin + 7 * kw)
    x = torch.randn(B, T, 1, Cin, generator=g).cuda()
    w = (torch.randn(kw, Cin, Cout, generator=g) / (Cin * kw) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    To = T + padl + padr - kw + 1
    dy = torch.randn(B, To, 1, Cout, generator=g).cuda()

    def ref(xx, ww, dd):
        xx = xx.double().reshape(B, T, Cin).permute(0, 2, 1)                 # [B][Cin][T]
        xx = F.pad(xx, (padl, padr)).requires_grad_(True)
        wt = ww.double().permute(2, 1, 0).contiguous().requires_grad_(True)  # [Cout][Cin][kw]
        y = F.conv1d(xx, wt)                                                 # [B][Cout][To]
        (y * dd.double().reshape(B, To, Cout).permute(0, 2, 1)).sum().backward()
        dx = xx.grad[:, :, padl:padl + T].permute(0, 2, 1).reshape(B, T, 1, Cin)
        return y.permute(0, 2, 1).reshape(B, To, 1, Cout).detach(), dx, wt.grad.permute(2, 1, 0)
    r = lambda t: t.bfloat16().float()
    L = _lib.lib()
    prev = L.w2l_set_matmul_precision(1)
    try:
        y = ops.conv_forward(x, w, bias, 1, padl, padr)
        dx, dw, db = ops.conv_backward(x, w, dy, 1, padl, padr)
    finally:
        L.w2l_set_matmul_precision(prev)
    yr, _, _ = ref(r(x), r(w), dy)
    assert rel(y, (yr + bias.double()).cpu().numpy()) < 2e-5
    _, dxr, _ = ref(x, r(w), r(dy))
    assert rel(dx, dxr.cpu().numpy()) < 2e-5
    y0, dx0, dw0 = ref(x, w, dy)
    assert rel(y, (y0 + bias.double()).cpu().numpy()) < BF16_TOL and rel(dx, dx0.cpu().numpy()) < BF16_TOL
    assert rel(dw, dw0.cpu().numpy()) < 1e-4
    assert rel(db, dy.double().sum((0, 1, 2)).cpu().numpy()) < 1e-4
    # and the fp32 mode is untouched
    assert rel(ops.conv_forward(x, w, bias, 1, padl, padr), (y0 + bias.double()).cpu().numpy()) < 1e-4


@pytest.mark.parametrize("n,M,N,K", [(4, 1024, 1024, 3008), (3, 3008, 1024, 1024), (2, 200, 136, 70), (1, 129, 260, 64), (4, 64, 64, 4100)])
def test_gemm_bf16_grouped(n, M, N, K):
    """w2l_gemm_bf16_grouped: up to four products of one shape in one launch (the Transformer block's four projection weight
    gradients / its q, k, v projections), each equal to the single launch bit for bit, with and without biases"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + M + K)
    As = [ops.bf16_convert(torch.randn(M, K, generator=g).cuda())[0] for _ in range(n)]
    Bs = [ops.bf16_convert((torch.randn(N, K, generator=g) / K ** 0.5).cuda())[0] for _ in range(n)]
    if n > 1:
        As[1] = As[0]                                     # shared operand (x images feed q, k and v)
    biases = [torch.randn(N, generator=g).cuda() if i != 1 else None for i in range(n)]
    outs = ops.gemm_bf16_grouped(As, Bs, K)
    for i in range(n):
        assert torch.equal(outs[i], ops.gemm_bf16(As[i], Bs[i], K))
    outs = ops.gemm_bf16_grouped(As, Bs, K, biases)
    for i in range(n):
        assert torch.equal(outs[i], ops.gemm_bf16(As[i], Bs[i], K, bias=biases[i]))
    with pytest.raises(Exception):
        ops.gemm_bf16_grouped((As * 5)[:5], (Bs * 5)[:5], K)   # more than four problems


@pytest.mark.parametrize("M,N,K", [(700, 520, 333), (4000, 3100, 333)])
def test_gemm_bf16_operand_storage_epilogues(oracle, M, N, K):
    """the fp32 engine's epilogue on the bf16 product: mask, addend, accumulate into C, dropout (the library's stateless hash:
    bit-identical keep pattern to w2l_dropout_inplace over the dense output)"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    mask = torch.randn(M, N, generator=g).cuda()
    add = torch.randn(M, N, generator=g).cuda()
    Ab, _ = ops.bf16_convert(A)
    Bb, _ = ops.bf16_convert(B)
    base = (_bf16_ref(A) @ _bf16_ref(B).T)
    got = ops.gemm_bf16(Ab, Bb, K, mask=mask, mask_scale=1.25)
    assert rel(got, (torch.where(mask.double() > 0, base * 1.25, torch.zeros_like(base))).cpu().numpy()) < 2e-5
    got = ops.gemm_bf16(Ab, Bb, K, addend=add)
    assert rel(got, (base + add.double()).cpu().numpy()) < 2e-5
    c = add.clone()
    ops.gemm_bf16(Ab, Bb, K, out=c, accumulate=True)
    assert rel(c, (base + add.double()).cpu().numpy()) < 2e-5
    p, seed, sid = 0.3, 1234, 7
    y = ops.gemm_bf16(Ab, Bb, K, bias=bias, relu=True, drop_p=p, drop_seed=seed, drop_stream=sid)
    y0 = ops.gemm_bf16(Ab, Bb, K, bias=bias, relu=True)
    ops.dropout_(y0, p, seed, sid)
    assert torch.equal(y, y0)


@pytest.mark.parametrize("M,N,K", [(700, 520, 333),       # 128 x 128 kernel, ragged last tile row (700 = 5 x 128 + 60) and column
                                   (4000, 3100, 333),     # many tiles; 3100 = 24 x 128 + 28 columns
                                   (11968, 1200, 400),    # config 3's frame count: the last tile row holds 64 rows
                                   (1024, 2048, 2048),    # 256 x 256 kernel
                                   (130, 36, 64)])        # a handful of rows / columns
def test_gemm_bf16_writes_the_bf16_images_of_its_result(M, N, K):
    """w2l_gemm_bf16_images: the result leaves as the two bf16 images w2l_bf16_convert would make of the fp32 C -- bit for bit, with
    bias + ReLU + dropout applied, padding untouched (zero) -- with or without the fp32 copy; and a bf16 image serves as the mask
    operand of a backward-data product exactly like the fp32 matrix it is the image of"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Ab, _ = ops.bf16_convert(A)
    Bb, _ = ops.bf16_convert(B)
    p, seed, sid = 0.2, 99, 5
    want = ops.gemm_bf16(Ab, Bb, K, bias=bias, relu=True, drop_p=p, drop_seed=seed, drop_stream=sid)
    wr, wt = ops.bf16_convert(want, True, True)
    for keep in (False, True):
        c, rm, tr = ops.gemm_bf16_images(Ab, Bb, K, bias=bias, relu=True, keep_f32=keep, drop_p=p, drop_seed=seed, drop_stream=sid)
        assert torch.equal(rm.view(torch.int16), wr.view(torch.int16))
        assert torch.equal(tr.view(torch.int16), wt.view(torch.int16))
        assert (c is None) if not keep else torch.equal(c, want)
    _, rm, _ = ops.gemm_bf16_images(Ab, Bb, K, bias=bias, relu=True, transposed_image=False, drop_p=p, drop_seed=seed, drop_stream=sid)
    assert torch.equal(rm.view(torch.int16), wr.view(torch.int16))
    _, _, tr = ops.gemm_bf16_images(Ab, Bb, K, bias=bias, relu=True, rows_image=False, drop_p=p, drop_seed=seed, drop_stream=sid)
    assert torch.equal(tr.view(torch.int16), wt.view(torch.int16))
    # the mask operand as a bf16 image: dX[M][N] = dY W^T masked by (u > 0), u given as its row image
    u = torch.randn(M, N, generator=g).cuda().clamp_min(0)      # a ReLU output: zeros and positives
    uImg, _ = ops.bf16_convert(u)
    ref = ops.gemm_bf16(Ab, Bb, K, mask=u, mask_scale=1.25)
    got, _, _ = ops.gemm_bf16_images(Ab, Bb, K, keep_f32=True, rows_image=False, transposed_image=False, mask_image=uImg, mask_scale=1.25)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,C,H,T,kw,padl,padr", [(2, 15, 80, 187, 9, 7, 1), (3, 19, 80, 70, 9, 7, 1), (2, 23, 80, 93, 11, 9, 1),
                                                  (2, 27, 80, 187, 11, 10, 0), (1, 27, 16, 5, 11, 10, 0), (2, 10, 80, 130, 21, 10, 10),
                                                  (1, 18, 80, 66, 21, 10, 10), (2, 14, 32, 64, 21, 10, 10)])
def test_tds_conv_bf16_three_passes(oracle, B, C, H, T, kw, padl, padr):
    """the TDS convolution of the mixed-precision mode (conv_tds_bf16.hip) at the streaming recipe's channel counts /
    kernel widths / asymmetric paddings (and the sota/2019 TDS-CTC ones): forward (+ bias, ReLU), backward-data (+ addend)
    and backward-filter against the oracle on the SAME bf16-rounded operands -- only the fp32 accumulation differs (2e-5 of the
    largest magnitude) -- plus the stated 1e-2 against the unrounded convolution and run-to-run determinism"""
    from oracle import refnet
    from wav2letter_amd import ops
    rng = np.random.default_rng(C * 100 + T)
    x = rng.normal(size=(B, C, H, T)).astype(np.float32)
    w = (rng.normal(size=(C, C, kw)) / np.sqrt(C * kw)).astype(np.float32)
    b = rng.normal(size=C).astype(np.float32)
    xr, wr = refnet.bf16_round(x), refnet.bf16_round(w)
    y_ref = oracle.conv_fwd(xr, wr, b, 1, padl, padr)
    xd, wd = dev(to_fm(x)), dev(w_to_dev(w))
    out = ops.tds_conv_bf16(xd, wd, dev(b), padl, padr)
    assert out is not None, "geometry of the recipes must have a bf16 kernel"
    y, imgs, d = out
    assert rel(from_fm(y.cpu().numpy()), y_ref) < 2e-5
    assert rel(from_fm(y.cpu().numpy()), oracle.conv_fwd(x, w, b, 1, padl, padr)) < BF16_TOL
    yr, _, _ = ops.tds_conv_bf16(xd, wd, dev(b), padl, padr, relu=True)
    assert rel(from_fm(yr.cpu().numpy()), np.maximum(y_ref, 0)) < 2e-5
    dy = rng.normal(size=y_ref.shape).astype(np.float32)
    add = rng.normal(size=x.shape).astype(np.float32)
    odx, _, _ = oracle.conv_bwd(xr, wr, refnet.bf16_round(dy), 1, padl, padr)
    _, odw, _ = oracle.conv_bwd(xr, wr, refnet.bf16_round(dy), 1, padl, padr)
    dyd, addd = dev(to_fm(dy)), dev(to_fm(add))
    dx, dw = ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd)
    assert rel(from_fm(dx.cpu().numpy()), odx + add) < 2e-5
    assert rel(dw.cpu().numpy(), w_to_dev(odw)) < 5e-5
    dx2, dw2 = ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)
    dx0, _ = ops.tds_conv_bf16_backward(xd, dyd, imgs, d)
    assert rel(from_fm(dx0.cpu().numpy()), odx) < 2e-5


@pytest.mark.parametrize("seed", [1, 2])
def test_tds_conv_bf16_random_geometries(oracle, seed):
    """the test above over random batch / frame / mel-row counts, kernel widths and paddings (geometries without a bf16 kernel are
    refused by the library -- `None` -- and skipped here: at least a third of the draws must run)"""
    rng = np.random.default_rng(31 + seed)
    ran = 0
    for c in range(20):
        B = int(rng.integers(1, 4))
        C = int(rng.choice([10, 14, 15, 18, 19, 23, 27]))
        H = int(rng.choice([16, 32, 80, 80, 24, 5]))
        T = int(rng.choice([1, 2, 5, 16, 17, 31, 33, 64, 65, 100, 187, 200]))
        kw = int(rng.choice([9, 10, 11, 12, 21]))
        padl, padr = int(rng.integers(0, kw)), int(rng.integers(0, kw))
        if T + padl + padr - kw + 1 < 1:
            padl = padr = kw - 1
        try:
            test_tds_conv_bf16_three_passes(oracle, B, C, H, T, kw, padl, padr)
            ran += 1
        except AssertionError as e:
            if "must have a bf16 kernel" in str(e):
                continue
            raise AssertionError(f"case {c}: B={B} C={C} H={H} T={T} kw={kw} padl={padl} padr={padr}: {e}") from e
    assert ran >= 6, ran


@pytest.mark.parametrize("rows,cols,p", [(300, 200, 0.1), (64, 64, 0.5), (1000, 1203, 0.25), (7, 5, 0.3), (4097, 130, 0.1)])
def test_bf16_convert_with_dropout_mask(rows, cols, p):
    """w2l_bf16_convert_dropout: the images of dropout(x) in one pass -- bit-identical to w2l_dropout_copy (same p, seed, stream:
    the library's stateless hash of the flat index) followed by w2l_bf16_convert, padding included; p = 0 is the plain conversion"""
    from wav2letter_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).cuda()
    want_r, want_t = ops.bf16_convert(ops.dropout_copy(x, p, 77, 5), True, True)
    got_r, got_t = ops.bf16_convert_dropout(x, p, 77, 5)
    assert torch.equal(got_r.view(torch.int16), want_r.view(torch.int16))
    assert torch.equal(got_t.view(torch.int16), want_t.view(torch.int16))
    zero = (want_r[:, :cols].float() == 0).float().mean().item()
    assert abs(zero - p) < 0.05 + 3.0 / (rows * cols) ** 0.5
    plain_r, plain_t = ops.bf16_convert(x, True, True)
    g0_r, g0_t = ops.bf16_convert_dropout(x, 0.0, 77, 5)
    assert torch.equal(g0_r.view(torch.int16), plain_r.view(torch.int16)) and torch.equal(g0_t.view(torch.int16), plain_t.view(torch.int16))


@pytest.mark.parametrize("groups,inner", [(1500, 1200), (37, 2160), (50, 1024), (3, 2304), (17, 8), (16, 64),
                                          (20, 1520), (21, 1840), (9, 300), (33, 1284)])   # the other chunks-per-lane instances
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_layernorm_writes_the_bf16_images_of_its_result(groups, inner, p):
    """w2l_residual_layernorm_forward_images / w2l_layernorm_backward_images (layernorm_images.hip: 16 rows per workgroup, the
    rounded rows through an LDS tile into the transposed image): r / y / dr / dmask / the parameter gradients against the plain
    kernels (the wave-level reduction order differs: 2e-6), the dropout pattern bit for bit, the images bit for bit the rounding
    of the kernel's OWN fp32 result -- of dropout(dr) with the library hash where asked (= w2l_bf16_convert_dropout) -- and nothing
    written beyond the matrix except zeros in the transposed image's row padding; rows the kernel does not hold are refused"""
    import ctypes as C
    from wav2letter_amd import _lib, ops
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(groups * 7 + inner)
    a = torch.relu(torch.randn(groups, inner, generator=g)).cuda()
    x = torch.randn(groups, inner, generator=g).cuda()
    gb = torch.tensor([1.3, -0.2]).cuda()
    a_ref = a.clone()
    y_ref, r_ref, mr_ref = ops.residual_layernorm_forward(a_ref, x, gb, groups, 1e-5, p, 77, 5)
    ldR, ldT = (inner + 63) // 64 * 64 + 64, (groups + 63) // 64 * 64

    def sink():
        rows = torch.full((groups, ldR), 7.0, dtype=torch.bfloat16, device="cuda")
        trans = torch.full((inner, ldT), 7.0, dtype=torch.bfloat16, device="cuda")
        return rows, trans, _lib.Bf16ImageSink(rowMajor=rows.data_ptr(), ldRows=ldR, transposed=trans.data_ptr(), ldTrans=ldT)

    def check_images(rows, trans, want32):
        want = want32.bfloat16()
        assert torch.equal(rows[:, :inner], want) and torch.equal(trans[:, :groups], want.t())
        assert bool((rows[:, inner:] == 7.0).all())
        g16 = (groups + 15) // 16 * 16
        assert bool((trans[:, groups:g16] == 0.0).all()) and bool((trans[:, g16:] == 7.0).all())

    a2 = a.clone()
    r = torch.empty_like(a); y = torch.empty_like(a); mr = torch.empty(2 * groups, device="cuda")
    rows, trans, k = sink()
    assert L.w2l_residual_layernorm_forward_images(groups, inner, a2.data_ptr(), x.data_ptr(), r.data_ptr(), y.data_ptr(), gb.data_ptr(), 1e-5, p,
                                                   77, 5, mr.data_ptr(), C.byref(k), s) == 0
    assert torch.equal(a2, a_ref) and torch.equal(r, r_ref)
    assert rel(y, y_ref.cpu().numpy()) < 2e-6 and rel(mr, mr_ref.cpu().numpy()) < 2e-6
    check_images(rows, trans, y)
    # r stored over a, no residual input: the Transformer block's call pattern
    a3 = a.clone()
    y3 = torch.empty_like(a)
    rows3, trans3, k3 = sink()
    assert L.w2l_residual_layernorm_forward_images(groups, inner, a3.data_ptr(), None, a3.data_ptr(), y3.data_ptr(), gb.data_ptr(), 1e-5, 0.0, 0, 0,
                                                   mr.data_ptr(), C.byref(k3), s) == 0
    y3_ref, _, _ = ops.residual_layernorm_forward(a.clone(), None, gb, groups)
    assert torch.equal(a3, a) and rel(y3, y3_ref.cpu().numpy()) < 2e-6
    check_images(rows3, trans3, y3)
    # backward
    dy = torch.randn(groups, inner, generator=g).cuda()
    for masked in (False, True):
        dr_ref, dgb_ref, dm_ref = ops.layernorm_backward(r_ref, dy, gb, mr_ref, groups, mask_src=a_ref if masked else None, mask_scale=1.25)
        dr = torch.empty_like(a); dm = torch.empty_like(a); dgb = torch.empty(2, device="cuda")
        sums = torch.empty(2 * groups + 64, dtype=torch.float64, device="cuda")
        rows, trans, k = sink()
        assert L.w2l_layernorm_backward_images(groups, inner, r_ref.data_ptr(), dy.data_ptr(), gb.data_ptr(), mr_ref.data_ptr(), dr.data_ptr(),
                                               dgb.data_ptr(), a_ref.data_ptr() if masked else None, dm.data_ptr() if masked else None, 1.25,
                                               sums.data_ptr(), C.byref(k), p, 91, 6, s) == 0
        assert rel(dr, dr_ref.cpu().numpy()) < 2e-6
        assert rel(dgb, dgb_ref.cpu().numpy()) < 1e-5
        if masked:
            assert rel(dm, dm_ref.cpu().numpy()) < 2e-6
        check_images(rows, trans, ops.dropout_copy(dr, p, 91, 6) if p > 0 else dr)
        if p > 0:
            want_r, want_t = ops.bf16_convert_dropout(dr, p, 91, 6)
            assert torch.equal(rows[:, :inner], want_r[:, :inner]) and torch.equal(trans[:, :groups], want_t[:inner, :groups])
    # refusals: a row the kernel does not hold, no images, a transposed pitch shorter than the rows rounded up to 16
    big = torch.zeros(2, 2308, device="cuda")
    assert L.w2l_residual_layernorm_forward_images(2, 2308, big.data_ptr(), None, big.data_ptr(), big.data_ptr(), gb.data_ptr(), 1e-5, 0.0, 0, 0,
                                                   mr.data_ptr(), C.byref(k), s) == _lib.W2L_EUNSUPPORTED
    none = _lib.Bf16ImageSink(rowMajor=None, ldRows=0, transposed=None, ldTrans=0)
    assert L.w2l_residual_layernorm_forward_images(groups, inner, a2.data_ptr(), None, a2.data_ptr(), y.data_ptr(), gb.data_ptr(), 1e-5, 0.0, 0, 0,
                                                   mr.data_ptr(), C.byref(none), s) == _lib.W2L_EINVAL
    short = _lib.Bf16ImageSink(rowMajor=None, ldRows=0, transposed=trans.data_ptr(), ldTrans=max(8, (groups - 1) // 8 * 8))
    if short.ldTrans < (groups + 15) // 16 * 16:
        assert L.w2l_residual_layernorm_forward_images(groups, inner, a2.data_ptr(), None, a2.data_ptr(), y.data_ptr(), gb.data_ptr(), 1e-5, 0.0, 0,
                                                       0, mr.data_ptr(), C.byref(short), s) == _lib.W2L_EINVAL


def test_tds_conv_bf16_refuses_other_geometries():
    import ctypes as C
    from wav2letter_amd import _lib, ops
    x = torch.zeros(1, 20, 80, 40, device="cuda")
    assert ops.tds_conv_bf16(x, torch.zeros(9, 40, 40, device="cuda"), None, 4, 4) is None          # C > 32
    x = torch.zeros(1, 20, 80, 15, device="cuda")
    assert ops.tds_conv_bf16(x, torch.zeros(7, 15, 15, device="cuda"), None, 3, 3) is None          # kw outside the instantiated set
    d = ops.conv_desc(x, torch.zeros(9, 15, 15, device="cuda"), 3, 4, 4)
    assert _lib.lib().w2l_tds_conv_bf16_image_elems(C.byref(d)) == 0                               # stride 3
    d = ops.conv_desc(x, torch.zeros(9, 15, 15, device="cuda"), 2, 4, 4)
    assert _lib.lib().w2l_tds_conv_bf16_image_elems(C.byref(d)) == 0                               # stride 2 at a kw without a kernel
    y = torch.zeros(1, 20, 80, 15, device="cuda")
    assert _lib.lib().w2l_tds_conv_bf16_forward(C.byref(d), x.data_ptr(), x.data_ptr(), None, y.data_ptr(), 0, None) == _lib.W2L_EUNSUPPORTED


@pytest.mark.parametrize("B,Cin,Cout,H,T,kw,stride,padl,padr", [
    (2, 1, 15, 80, 301, 10, 2, 5, 3), (2, 15, 19, 80, 160, 10, 2, 7, 1), (2, 19, 23, 80, 131, 12, 2, 9, 1), (2, 23, 27, 80, 70, 11, 1, 10, 0),
    (1, 19, 23, 16, 13, 12, 2, 9, 1), (2, 1, 10, 80, 200, 21, 2, 10, 10), (2, 10, 14, 80, 171, 21, 2, 10, 10), (1, 14, 18, 32, 90, 21, 2, 10, 10),
    (1, 15, 19, 16, 64, 10, 2, 0, 0)])
def test_subsampling_conv_bf16_three_passes(oracle, B, Cin, Cout, H, T, kw, stride, padl, padr):
    """the recipes' SUB-SAMPLING convolutions (C_in != C_out, stride 2 or 1; streaming recipe 1 -> 15 -> 19 -> 23 -> 27 with its
    asymmetric PD paddings, sota/2019 TDS-CTC 1 -> 10 -> 14 -> 18 at kw 21) on the bf16 kernels: forward (+ bias, ReLU), the
    phase-decomposed backward-data (+ addend; odd and even lengths) and backward-filter against the oracle on the SAME
    bf16-rounded operands (fp32 accumulation order is the only difference: 2e-5 / 5e-5 of the largest magnitude), the stated
    1e-2 against the unrounded convolution, run-to-run determinism"""
    from oracle import refnet
    from wav2letter_amd import ops
    rng = np.random.default_rng(Cin * 100 + T)
    x = rng.normal(size=(B, Cin, H, T)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, kw)) / np.sqrt(Cin * kw)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    xr, wr = refnet.bf16_round(x), refnet.bf16_round(w)
    y_ref = oracle.conv_fwd(xr, wr, b, stride, padl, padr)
    xd, wd = dev(to_fm(x)), dev(w_to_dev(w))
    out = ops.tds_conv_bf16(xd, wd, dev(b), padl, padr, stride=stride)
    assert out is not None, "geometry of the recipes must have a bf16 kernel"
    y, imgs, d = out
    assert rel(from_fm(y.cpu().numpy()), y_ref) < 2e-5
    assert rel(from_fm(y.cpu().numpy()), oracle.conv_fwd(x, w, b, stride, padl, padr)) < BF16_TOL
    yr, _, _ = ops.tds_conv_bf16(xd, wd, dev(b), padl, padr, relu=True, stride=stride)
    assert rel(from_fm(yr.cpu().numpy()), np.maximum(y_ref, 0)) < 2e-5
    dy = rng.normal(size=y_ref.shape).astype(np.float32)
    add = rng.normal(size=x.shape).astype(np.float32)
    odx, odw, odb = oracle.conv_bwd(xr, wr, refnet.bf16_round(dy), stride, padl, padr)
    dyd, addd = dev(to_fm(dy)), dev(to_fm(add))
    dx, dw = ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd)
    assert rel(from_fm(dx.cpu().numpy()), odx + add) < 2e-5
    assert rel(dw.cpu().numpy(), w_to_dev(odw)) < 5e-5
    # the bias gradient from the same launch: column sums of the rounded dy; dx / dw unchanged by the option
    dxb, dwb, db = ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd, with_bias=True)
    assert torch.equal(dxb, dx) and torch.equal(dwb, dw)
    assert rel(db.cpu().numpy(), odb) < 1e-5
    assert torch.equal(db, ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd, with_bias=True)[2])
    dx2, dw2 = ops.tds_conv_bf16_backward(xd, dyd, imgs, d, add=addd)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)
    dx0, _ = ops.tds_conv_bf16_backward(xd, dyd, imgs, d)
    assert rel(from_fm(dx0.cpu().numpy()), odx) < 2e-5
