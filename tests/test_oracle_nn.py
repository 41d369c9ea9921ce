"""Pin the CPU oracle for the network operators: the reference's two golden
vectors (Conv1dTest.cpp:30-104, TDSBlockTest.cpp:27-188) and torch-CPU
cross-checks of every forward/backward."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import refnet

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_conv1d_streaming_form(oracle):
    g = json.load(open(os.path.join(GOLD, "conv1d_golden.json")))
    T, G, Cc = g["T"], g["groups"], g["channels"] // g["groups"]
    y = oracle.streaming_conv1d(np.array(g["input"]), np.array(g["weights"]), np.array(g["bias"]),
                                T, G, Cc, Cc, g["kernelSize"], g["stride"], g["leftPadding"], g["rightPadding"])
    err = np.abs(y - np.array(g["target"])).max()
    assert err < g["tol"], err
    assert err < 2e-3  # fp16-packed weights in the reference => ~1e-3


def test_golden_conv1d_train_form(oracle):
    """same vector through the train-time Conv2D restatement ([B][C][H][T], w [Cout][Cin][kw])"""
    g = json.load(open(os.path.join(GOLD, "conv1d_golden.json")))
    T, H, Cc, kw = g["T"], g["groups"], g["channels"] // g["groups"], g["kernelSize"]
    x = np.array(g["input"], np.float32).reshape(T, H, Cc).transpose(2, 1, 0)[None]  # [1][C][H][T]
    w = np.array(g["weights"], np.float32).reshape(Cc, kw, Cc).transpose(0, 2, 1)     # [co][ci][k]
    y = oracle.conv_fwd(x, w, np.array(g["bias"], np.float32), 1, g["leftPadding"], g["rightPadding"])
    got = y[0].transpose(2, 1, 0).reshape(-1)
    assert np.abs(got - np.array(g["target"])).max() < 2e-3


def test_golden_tdsblock(oracle):
    g = json.load(open(os.path.join(GOLD, "tdsblock_golden.json")))
    T, H, Cc, kw = g["T"], g["groups"], g["channels"] // g["groups"], g["kernelSize"]
    p = refnet.TDSParams(Cc, kw, H)
    p.wc = np.array(g["conv_weights"], np.float32).reshape(Cc, kw, Cc).transpose(0, 2, 1).copy()
    p.bc = np.array(g["conv_bias"], np.float32)
    p.g1, p.b1n = np.float32(g["ln1_weights"][0]), np.float32(g["ln1_bias"][0])
    p.w1 = np.array(g["lin1_weights"], np.float32).reshape(10, 10)  # memory [in][out]
    p.b1 = np.array(g["lin1_bias"], np.float32)
    p.w2 = np.array(g["lin2_weights"], np.float32).reshape(10, 10)
    p.b2 = np.array(g["lin2_bias"], np.float32)
    p.g2, p.b2n = np.float32(g["ln2_weights"][0]), np.float32(g["ln2_bias"][0])
    x = np.array(g["in"], np.float32).reshape(T, H, Cc).transpose(2, 1, 0)[None].copy()
    out = refnet.tds_fwd(x, p, g["leftPadding"], g["rightPadding"], ln_mode="frame", streaming=True)
    got = out[0].transpose(2, 1, 0).reshape(-1)
    err = np.abs(got - np.array(g["expectedOutput"])).max()
    assert err < g["tol"], err
    assert err < 2e-3
    # train-time LayerNorm (eps inside sqrt) differs only at the 1e-4 level here
    out2 = refnet.tds_fwd(x, p, 1, 1, ln_mode="frame", streaming=False)
    assert np.abs(out2 - out).max() < 5e-3


@pytest.mark.parametrize("stride,kw,padl,padr,H", [(1, 5, 2, 2, 3), (2, 7, 3, 3, 2), (1, 4, 0, 0, 1), (2, 21, 10, 10, 4)])
def test_conv_vs_torch(oracle, stride, kw, padl, padr, H):
    rng = np.random.default_rng(kw)
    B, Cin, Cout, T = 2, 3, 5, 23
    x = rng.normal(size=(B, Cin, H, T)).astype(np.float32)
    w = rng.normal(size=(Cout, Cin, kw)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    xp = F.pad(xt, (padl, padr))
    yt = F.conv2d(xp, wt[:, :, None, :], bt, stride=(1, stride))
    y = oracle.conv_fwd(x, w, b, stride, padl, padr)
    assert y.shape == tuple(yt.shape)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-5)
    dy = rng.normal(size=y.shape).astype(np.float32)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    dx, dw, db = oracle.conv_bwd(x, w, dy, stride, padl, padr)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-5)
    assert np.allclose(dw, wt.grad.numpy(), atol=1e-4)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-4)


def test_same_pad_semantics(oracle):
    # odd kw stride 1 => T'=T; even kw stride 1 => T'=T+1; kw=21 stride 2 => ceil(T/2)
    for T in (100, 101, 1500, 750, 375):
        p = oracle.same_pad(T, 21, 2)
        assert oracle.conv_out_len(T, 21, 2, p, p) == (T + 1) // 2
    p = oracle.same_pad(50, 7, 1)
    assert oracle.conv_out_len(50, 7, 1, p, p) == 50
    p = oracle.same_pad(50, 4, 1)
    assert oracle.conv_out_len(50, 4, 1, p, p) == 51


def test_linear_vs_torch(oracle):
    rng = np.random.default_rng(1)
    M, K, N = 7, 5, 9
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = rng.normal(size=(K, N)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    assert np.allclose(oracle.linear_fwd(x, w, b), x.astype(np.float64) @ w + b, atol=1e-5)
    dy = rng.normal(size=(M, N)).astype(np.float32)
    dx, dw, db = oracle.linear_bwd(x, w, dy)
    assert np.allclose(dx, dy.astype(np.float64) @ w.T, atol=1e-5)
    assert np.allclose(dw, x.T.astype(np.float64) @ dy, atol=1e-5)
    assert np.allclose(db, dy.sum(0), atol=1e-5)


def test_layernorm_vs_torch(oracle):
    rng = np.random.default_rng(2)
    G, inner = 3, 40
    x = rng.normal(size=(G, inner)).astype(np.float32) * 2 + 1
    dy = rng.normal(size=(G, inner)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gam = torch.tensor(1.3, dtype=torch.float64, requires_grad=True)
    bet = torch.tensor(-0.2, dtype=torch.float64, requires_grad=True)
    yt = F.layer_norm(xt, (inner,), eps=1e-5) * gam + bet
    y = oracle.layernorm_fwd(x, G, 1.3, -0.2, 1e-5)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-5)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    dx, dg, db = oracle.layernorm_bwd(x, dy, G, 1.3, 1e-5)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-5)
    assert abs(dg - gam.grad.item()) < 1e-4 and abs(db - bet.grad.item()) < 1e-4


def test_glu_weightnorm_vs_torch(oracle):
    rng = np.random.default_rng(3)
    outer, half, inner = 2, 3, 5
    x = rng.normal(size=(outer, 2 * half, inner)).astype(np.float32)
    dy = rng.normal(size=(outer, half, inner)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = F.glu(xt, 1)
    assert np.allclose(oracle.glu_fwd(x, outer, half, inner).reshape(yt.shape), yt.detach().numpy(), atol=1e-6)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    assert np.allclose(oracle.glu_bwd(x, dy, outer, half, inner), xt.grad.numpy(), atol=1e-6)
    # WN dim 3 on conv weights [Cout][Cin][kw]
    v = rng.normal(size=(4, 3, 5)).astype(np.float32)
    g = rng.normal(size=4).astype(np.float32)
    dw = rng.normal(size=v.shape).astype(np.float32)
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    wt = vt * (gt / vt.reshape(4, -1).norm(dim=1))[:, None, None]
    assert np.allclose(oracle.weightnorm_fwd(v, g, 1, 4, 15), wt.detach().numpy(), atol=1e-6)
    wt.backward(torch.tensor(dw, dtype=torch.float64))
    dv, dg = oracle.weightnorm_bwd(v, g, dw, 1, 4, 15)
    assert np.allclose(dv, vt.grad.numpy(), atol=1e-5) and np.allclose(dg, gt.grad.numpy(), atol=1e-5)
    # WN dim 0 on linear W memory [in][out]: norm over `in` per output column
    v = rng.normal(size=(6, 4)).astype(np.float32)
    w = oracle.weightnorm_fwd(v, g, 6, 4, 1)
    assert np.allclose(w, v * (g / np.linalg.norm(v.astype(np.float64), axis=0)), atol=1e-6)


@pytest.mark.parametrize("ln_mode", ["all", "frame"])
def test_tds_block_backward_vs_torch(oracle, ln_mode):
    rng = np.random.default_rng(4)
    B, Cc, H, T, kw = 2, 3, 4, 9, 5
    p = refnet.TDSParams(Cc, kw, H, l2=20, rng=rng)
    x = rng.normal(size=(B, Cc, H, T)).astype(np.float32)
    dout = rng.normal(size=x.shape).astype(np.float32)
    out, saved = refnet.tds_fwd(x, p, 2, 2, ln_mode, keep=True)
    dx, g = refnet.tds_bwd(dout, p, saved, 2, 2, ln_mode)

    td = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)
    xt, wc, bc, w1, b1, w2, b2 = map(td, (x, p.wc, p.bc, p.w1, p.b1, p.w2, p.b2))
    g1, b1n, g2, b2n = map(td, (p.g1, p.b1n, p.g2, p.b2n))

    def ln(v, gam, bet):
        dims = (1, 2, 3) if ln_mode == "all" else (1, 2)
        mu = v.mean(dims, keepdim=True)
        var = ((v - mu) ** 2).mean(dims, keepdim=True)
        return (v - mu) / torch.sqrt(var + 1e-5) * gam + bet

    a = F.conv2d(F.pad(xt, (2, 2)), wc[:, :, None, :], bc)
    y = ln(torch.relu(a) + xt, g1, b1n)
    z = y.permute(0, 3, 2, 1).reshape(B * T, H * Cc)
    v = torch.relu(z @ w1 + b1) @ w2 + b2
    s = v.reshape(B, T, H, Cc).permute(0, 3, 2, 1) + y
    ot = ln(s, g2, b2n)
    assert np.allclose(out, ot.detach().numpy(), atol=1e-4)
    ot.backward(torch.tensor(dout, dtype=torch.float64))
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-4)
    for name, t in dict(wc=wc, bc=bc, w1=w1, b1=b1, w2=w2, b2=b2, g1=g1, b1n=b1n, g2=g2, b2n=b2n).items():
        assert np.allclose(g[name], t.grad.numpy(), atol=2e-4), name


def test_dropout_hash_statistics(oracle):
    x = np.ones(200000, np.float32)
    y = oracle.dropout(x, 0.25, 1234, 7)
    keep = (y != 0).mean()
    assert abs(keep - 0.75) < 0.01
    assert np.allclose(y[y != 0], 1 / 0.75)
    y2 = oracle.dropout(x, 0.25, 1234, 8)
    assert 0.5 < ((y != 0) == (y2 != 0)).mean() < 0.7  # streams decorrelated (0.75^2+0.25^2=0.625)


def test_mfsc_oracle_properties(oracle):
    """MFSC restatement (parity unpinned: the arithmetic is un-vendored Flashlight).  What the reference's own test
    checks (LogMelFeatureTest.cpp:25-66): the features do not depend on how the audio is cut into chunks -- frame t
    only sees samples [160 t, 160 t + 400).  Plus: shape conventions, the filterbank covers every FFT bin below
    Nyquist with non-negative triangles, and the folded linear form (pre-emphasis, window, DFT as one matrix) that the
    device path multiplies by equals the frame-by-frame computation."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=16000) * 2000.0
    full = oracle.mfsc(x, 80)
    assert full.shape == (1 + (16000 - 400) // 160, 80)
    for cut in (400, 1999, 8000, 12345):
        part = oracle.mfsc(x[:cut], 80)
        assert part.shape[0] == (0 if cut < 400 else 1 + (cut - 400) // 160)
        assert np.abs(part - full[:part.shape[0]]).max() < 1e-12
    assert oracle.mfsc(x[:399], 80).shape == (0, 80)
    H = oracle.mfsc_filterbank(40, 512, 16000)
    assert H.shape == (257, 40) and (H >= 0).all() and (H.sum(0) > 0).all() and (H[1:256].sum(1) > 0).all()
    N, S, nb = 400, 160, 257
    n = np.arange(N)
    win = 0.54 - 0.46 * np.cos(2 * np.pi * n / (N - 1))
    P = np.eye(N) - 0.97 * np.eye(N, k=-1)
    P[0, 0] = 0.03
    ang = 2 * np.pi * np.outer(n, np.arange(nb)) / 512
    WP = np.diag(win) @ P
    G = np.concatenate([WP.T @ np.cos(ang), -(WP.T @ np.sin(ang))], axis=1)
    T = full.shape[0]
    frames = np.stack([x[t * S:t * S + N] for t in range(T)])
    ri = frames @ G
    folded = np.log(np.maximum(np.sqrt(ri[:, :nb] ** 2 + ri[:, nb:] ** 2) @ oracle.mfsc_filterbank(80, 512, 16000), 1.0))
    assert np.abs(folded - full).max() < 1e-10
    assert np.abs(oracle.mfsc(x, 80, use_power=True) - oracle.mfsc(x, 80)).max() > 1.0   # power and magnitude differ


@pytest.mark.parametrize("T,F,args", [(1500, 80, (27, 2, 100, 1.0, 2)), (57, 40, (15, 1, 50, 0.2, 2)), (9, 8, (8, 2, 100, 1.0, 1))])
def test_specaugment_oracle_properties(oracle, T, F, args):
    """fl::SpecAugment restatement (un-vendored => parity unpinned): one mask set per batch, inclusive af::seq ends,
    widths drawn below fMaskF / min(tMaskT, T*p), whole rows / columns zeroed, everything else untouched"""
    fmf, nf, tmt, tmp_, nt = args
    rng = np.random.default_rng(T)
    x = rng.normal(size=(3, T, F)).astype(np.float32)
    x[x == 0] = 1.0
    outs = set()
    for seed in range(1, 20):
        y, m = oracle.specaugment(x, fmf, nf, tmt, tmp_, nt, seed)
        zero = y == 0
        assert (zero == zero[0:1]).all()
        fm, tm = zero[0].all(axis=0), zero[0].all(axis=1)
        assert (zero[0] == (fm[None, :] | tm[:, None])).all()
        assert (y[~zero] == x[~zero]).all()
        tmax = min(tmt, int(T * tmp_), T)
        want_f = np.zeros(F, bool)
        want_t = np.zeros(T, bool)
        for k in range(nf):
            assert 0 <= m[0, k] <= m[1, k] < F and m[1, k] - m[0, k] < fmf
            want_f[m[0, k]:m[1, k] + 1] = True
        for k in range(nt):
            assert 0 <= m[2, k] <= m[3, k] < T and m[3, k] - m[2, k] < tmax
            want_t[m[2, k]:m[3, k] + 1] = True
        assert (zero[0] == (want_f[None, :] | want_t[:, None])).all()
        outs.add(y.tobytes())
    assert len(outs) > 10
    with pytest.raises(ValueError):
        oracle.specaugment(x[:, :, :4], 27, 2, 100, 1.0, 2, 1)


def test_torch_cpu_proxy_matches_reference_interpreter(oracle):
    """oracle/torchnet.py (the torch-CPU proxy timed as bench.py's cpu_baseline) computes the same network as
    oracle/refnet.RefNet: emissions and parameter gradients on a reduced TDS-CTC of the recipe's topology"""
    from oracle import torchnet
    from wav2letter_amd import recipes
    arch = recipes.tds_ctc_small_arch(c=(4, 6), h=8, kw=5)
    nfeat, nlabel, B, T = 8, 12, 2, 40
    rng = np.random.default_rng(3)
    ref = refnet.RefNet(arch, nfeat, nlabel)
    params = ref.random_params(rng)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    em = ref.forward(x, params)
    tp = [torch.from_numpy(p.copy()).requires_grad_(True) for p in params]
    tem = torchnet.TorchNet(arch, nfeat, nlabel).forward(torch.from_numpy(x), tp)
    assert tem.shape == em.shape
    assert np.abs(tem.detach().numpy() - em).max() < 1e-4 * np.abs(em).max()
    d = rng.normal(size=em.shape).astype(np.float32)
    g = ref.backward(d, len(params))
    tem.backward(torch.from_numpy(d))
    for a, b in zip(tp, g):
        b = np.asarray(b, np.float64)
        assert np.abs(a.grad.numpy() - b).max() < 2e-4 * max(1e-6, np.abs(b).max())
    med, times = torchnet.tds_ctc_step_seconds(arch, nfeat, nlabel, B, T, L=4, warmup=1, runs=2)
    assert med > 0 and len(times) == 2


def test_two_dimensional_conv_oracles_agree(oracle):
    """kh x kw Conv2D (am_tds_ctc_librivox.arch): the numpy / C reference network (mel axis unrolled into channels) against
    torch's conv2d on the [cout][cin][kh][kw] kernel -- emissions and every parameter gradient"""
    from oracle import torchnet
    arch = ("V -1 NFEAT 1 0\nC2 1 4 5 3 2 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\nTDS 4 5 8 0.0 48\nC2 4 6 5 3 1 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\n"
            "TDS 6 5 8 0.0 0\nV 0 48 1 0\nRO 1 0 3 2\nL 48 NLABEL\n")
    nfeat, nlabel, B, T = 8, 11, 2, 36
    rng = np.random.default_rng(8)
    ref = refnet.RefNet(arch, nfeat, nlabel)
    params = ref.random_params(rng)
    assert params[0].shape == (4, 1, 3, 5)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    em = ref.forward(x, params)
    tp = [torch.from_numpy(p.astype(np.float64)).requires_grad_(True) for p in params]
    tem = torchnet.TorchNet(arch, nfeat, nlabel).forward(torch.from_numpy(x.astype(np.float64)), tp)
    assert tem.shape == em.shape
    assert np.abs(tem.detach().numpy() - em).max() < 1e-4 * np.abs(em).max()
    d = rng.normal(size=em.shape).astype(np.float32)
    g = ref.backward(d, len(params))
    tem.backward(torch.from_numpy(d.astype(np.float64)))
    for a, b in zip(tp, g):
        b = np.asarray(b, np.float64)
        assert a.grad.shape == b.shape
        assert np.abs(a.grad.numpy() - b).max() < 2e-4 * max(1e-6, np.abs(b).max())


def test_transformer_oracle_relative_position_rotate_and_block():
    """oracle/transformer_oracle.py: (a) the literal restatement of relativePositionEmbeddingRotate (pad, re-pitch, slice)
    equals the closed form rel[i][j] = q_i . E[j - i + csz - 1], zero outside the table -- for tables longer and shorter
    than the utterance; (b) attention rows sum to one through the block's plumbing: with v = 1 and Wf = I the attention
    sublayer outputs ones; (c) the interpreter's `M` max-pool routes gradients to the first maximum"""
    import torch
    from oracle import transformer_oracle as TO
    torch.manual_seed(0)
    for T, csz in [(7, 12), (9, 4), (5, 5), (1, 3), (12, 1)]:
        d, d0 = 8, 2 * csz - 1
        q = torch.randn(2, 3, T, d, dtype=torch.float64)
        E = torch.randn(d0, d, dtype=torch.float64)
        rot = TO.relative_position_rotate(q @ E.t())[..., d0 // 2:d0 // 2 + T]
        want = torch.zeros(2, 3, T, T, dtype=torch.float64)
        for i in range(T):
            for j in range(T):
                r = j - i + csz - 1
                if 0 <= r < d0:
                    want[..., i, j] = q[..., i, :] @ E[r]
        assert (rot - want).abs().max() < 1e-12, (T, csz)
    B, T, C, H = 2, 6, 8, 2
    q = torch.randn(B, T, C, dtype=torch.float64)
    ctx = TO.attention(q, torch.randn(B, T, C, dtype=torch.float64), torch.ones(B, T, C, dtype=torch.float64),
                       torch.randn(9, C // H, dtype=torch.float64), H)
    assert (ctx - 1).abs().max() < 1e-12
    arch = "V -1 1 NFEAT 0\nM 3 1 2 1\nRO 2 0 3 1\nL 4 NLABEL\n"
    net = refnet.RefNet(arch, 4, 3)
    x = np.array([[1, 5, 5, 2, 0, 7, 7]], np.float32).repeat(4, 0).reshape(1, 1, 4, 7)
    ps = net.random_params(np.random.default_rng(0))
    em = net.forward(x, ps)
    assert em.shape == (1, 3, 3)
    net.backward(np.ones_like(em), len(ps))
    # windows [0..2], [2..4], [4..6]: first maxima at t = 1, 2, 5
    k, shp, arg, sx = net.tape[1]
    assert k == "M" and (arg[0, 0, 0] == [1, 0, 1]).all()


def test_transformer_oracle_padding_mask_key_lengths():
    """oracle/transformer_oracle.key_lengths: the longest utterance keeps every key, sizes scale linearly, an equal-length
    resize is the identity, and masked attention rows still sum to one over the valid keys only"""
    import torch
    from oracle import transformer_oracle as TO
    assert list(TO.key_lengths([10.0, 5.0, 2.5], 40, 40)) == [40, 20, 10]
    assert list(TO.key_lengths([3.0, 3.0], 1500, 188)) == [188, 188]
    kl = TO.key_lengths([16000.0, 8000.0, 100.0], 1500, 188)
    assert kl[0] == 188 and abs(kl[1] - 94) <= 1 and 1 <= kl[2] <= 2
    B, T, C, H = 3, 9, 8, 2
    g = torch.Generator().manual_seed(0)
    q, k = (torch.randn(B, T, C, generator=g, dtype=torch.float64) for _ in range(2))
    v = torch.zeros(B, T, C, dtype=torch.float64)
    v[:, :, 0] = torch.arange(T, dtype=torch.float64)            # ctx[..., 0] = expected key index
    ctx = TO.attention(q, k, v, None, H, key_len=[9, 4, 1])
    assert ctx[1, :, 0].max() <= 3.0 + 1e-12 and (ctx[2, :, 0].abs() < 1e-12).all() and ctx[0, :, 0].max() > 3.0


def test_transformer_oracle_attention_against_torch_sdpa():
    """independent pin of the restated attention core where a second implementation exists: without the position table and
    masks, oracle/transformer_oracle.attention is torch's scaled_dot_product_attention per (utterance, head); with key lengths
    it is SDPA under the boolean key-padding mask"""
    import math
    import torch
    import torch.nn.functional as F
    from oracle import transformer_oracle as TO
    g = torch.Generator().manual_seed(3)
    B, T, H, d = 3, 11, 4, 8
    q, k, v = (torch.randn(B, T, H * d, generator=g, dtype=torch.float64) for _ in range(3))
    split = lambda z: z.reshape(B, T, H, d).permute(0, 2, 1, 3)
    got = TO.attention(q / math.sqrt(d), k, v, None, H)
    want = F.scaled_dot_product_attention(split(q), split(k), split(v)).permute(0, 2, 1, 3).reshape(B, T, H * d)
    assert (got - want).abs().max() < 1e-12
    kl = [11, 6, 2]
    keep = (torch.arange(T)[None, :] < torch.tensor(kl)[:, None])[:, None, None, :]
    got = TO.attention(q / math.sqrt(d), k, v, None, H, key_len=kl)
    want = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=keep).permute(0, 2, 1, 3).reshape(B, T, H * d)
    assert (got - want).abs().max() < 1e-12


def test_transformer_block_golden_vector_oracle_side():
    """tests/golden/transformer_block_golden.json (the hand-over vector for a reference-side check; generated by
    tests/golden/make_transformer_golden.py from the oracle): the oracle reproduces it, through the reference-layout
    interpreter too"""
    import json
    import os
    import torch
    from oracle import transformer_oracle as TO
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "transformer_block_golden.json")))
    C, T, B = g["modelDim"], g["T"], g["B"]
    # LayerNorm entries hold the module's two scalar parameters (weight, bias), each of af dims (1)
    params = [np.array(p["data"], np.float32).reshape(p["af_dims"][::-1] if "norm" not in p["name"] else (2,)) for p in g["params"]]
    x = np.array(g["x"], np.float32).reshape(B, T, C)
    y = TO.tr_block(torch.tensor(x, dtype=torch.float64), [torch.tensor(p, dtype=torch.float64) for p in params],
                    g["nHeads"], g["csz"]).numpy()
    assert np.abs(y.reshape(-1) - np.array(g["y"])).max() < 1e-9
    assert list(TO.key_lengths(g["input_sizes"], T, T)) == g["key_lengths"]
    net = refnet.RefNet("V -1 1 NFEAT 0\nRO 2 0 3 1\n" + g["arch_line"] + "\n", C, C)
    em = net.forward(np.ascontiguousarray(x.transpose(0, 2, 1))[:, None], params)     # input (T, 1, C, B) == [B][1][C][T]
    assert np.abs(em.reshape(-1) - np.array(g["y"], np.float32)).max() < 1e-5
    net.input_sizes = np.array(g["input_sizes"], np.float32)
    em = net.forward(np.ascontiguousarray(x.transpose(0, 2, 1))[:, None], params)
    assert np.abs(em.reshape(-1) - np.array(g["y_masked"], np.float32)).max() < 1e-5


def test_streaming_arch_reference_interpreter_against_torch(oracle):
    """BASELINE config 3's arch (am_500ms_future_context.arch: `PD` asymmetric padding ahead of unpadded strided
    convolutions, per-frame LayerNorm, TDS blocks with a right padding and lNormIncludeTime = 0, the V / RO tail) through
    oracle/refnet.RefNet against torch autograd (oracle/torchnet.py): emissions and every parameter gradient.  Pins the
    reference side of tests/test_gpu_trainer.py::test_streaming_tds_config3_full_network_end_to_end."""
    import re
    from oracle import torchnet
    from wav2letter_amd import recipes
    arch = re.sub(r"(TDS \d+ \d+ \d+) [0-9.]+", r"\1 0.0", recipes.streaming_tds_arch())
    arch = "\n".join(l for l in arch.splitlines() if not l.startswith("SAUG")) + "\n"
    arch = re.sub(r"^DO [0-9.]+$", "DO 0.0", arch, flags=re.M)
    nfeat, nlabel, B, T = 80, 50, 2, 96
    rng = np.random.default_rng(8)
    ref = refnet.RefNet(arch, nfeat, nlabel)
    params = ref.random_params(rng)
    x = rng.normal(size=(B, 1, nfeat, T)).astype(np.float32)
    em = ref.forward(x, params)
    assert em.shape == (B, 12, nlabel)          # 96 -> 48 -> 24 -> 12 -> 12 frames
    tp = [torch.from_numpy(p.copy()).requires_grad_(True) for p in params]
    tem = torchnet.TorchNet(arch, nfeat, nlabel).forward(torch.from_numpy(x), tp)
    assert tem.shape == em.shape
    assert np.abs(tem.detach().numpy() - em).max() < 1e-4 * np.abs(em).max()
    d = rng.normal(size=em.shape).astype(np.float32)
    g = ref.backward(d, len(params))
    tem.backward(torch.from_numpy(d))
    for a, b in zip(tp, g):
        b = np.asarray(b, np.float64).reshape(a.shape)
        assert np.abs(a.grad.numpy() - b).max() < 5e-4 * max(1e-6, np.abs(b).max())


def test_bf16_operand_rounding_of_the_reference_interpreter(oracle):
    """refnet.bf16_round == torch's fp32 -> bfloat16 conversion (round to nearest even) bit for bit, incl. ties, and the
    bf16 mode of the reference interpreter multiplies exactly those rounded operands"""
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.normal(size=4096).astype(np.float32) * 10.0 ** rng.integers(-6, 6, 4096),
                        np.array([1.0, 1.00390625, 1.01171875, -1.00390625, 0.0, 3.0e38, 1e-39], np.float32)])
    want = torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    got = refnet.bf16_round(a)
    assert got.tobytes() == want.tobytes()
    x = rng.normal(size=(7, 24)).astype(np.float32); w = rng.normal(size=(24, 5)).astype(np.float32); b = rng.normal(size=5).astype(np.float32)
    y = refnet.lin_fwd(x, w, b, bf16=True)
    yt = (torch.from_numpy(refnet.bf16_round(x)).double() @ torch.from_numpy(refnet.bf16_round(w)).double() + torch.from_numpy(b).double()).numpy()
    assert np.abs(y - yt).max() < 1e-5 * np.abs(yt).max()
    dy = rng.normal(size=(7, 5)).astype(np.float32)
    dx, dw, db = refnet.lin_bwd(x, w, dy, bf16=True)
    assert np.abs(dx - refnet.bf16_round(dy).astype(np.float64) @ refnet.bf16_round(w).astype(np.float64).T).max() < 1e-5
    assert np.abs(dw - refnet.bf16_round(x).astype(np.float64).T @ refnet.bf16_round(dy).astype(np.float64)).max() < 1e-5
    assert np.abs(db - refnet.bf16_round(dy).astype(np.float64).sum(0)).max() < 1e-5   # the bias gradient rides on the weight-gradient product: sums of the ROUNDED dy


def test_oracle_and_library_agree_on_which_convolutions_multiply_in_bf16():
    """the mixed-precision mode rounds a convolution's operands to bf16 exactly when the library has bf16 kernels for its
    geometry (w2l_tds_conv_bf16_image_elems != 0: host logic, no GPU needed); the oracle's restatement of that rule
    (refnet.conv_rounds_to_bf16) must give the same answer for every geometry, or a full-network comparison would silently
    compare a rounded product with an unrounded reference"""
    import ctypes as C
    from oracle import refnet
    from wav2letter_amd import _lib
    L = _lib.lib()
    n = yes = 0
    for stride in (1, 2, 3):
        for H in (8, 16, 80):
            for kw in (1, 5, 9, 10, 11, 12, 21):
                for cin in (1, 3, 10, 14, 15, 16, 18, 19, 23, 24, 27, 32, 33):
                    for cout in (1, 10, 14, 15, 16, 18, 19, 23, 27, 32, 40):
                        for padl, padr in ((0, 0), (kw - 1, 0), (kw // 2, (kw - 1) // 2)):
                            d = _lib.ConvDesc(2, 64, H, cin, cout, kw, stride, padl, padr)
                            lib_says = L.w2l_tds_conv_bf16_image_elems(C.byref(d)) != 0
                            assert lib_says == refnet.conv_rounds_to_bf16(cin, cout, kw, stride, H), (cin, cout, kw, stride, H, padl, padr)
                            n += 1
                            yes += lib_says
    assert n > 10000 and yes > 50
    # the recipes' layers are among them
    for cin, cout, kw, stride in ((1, 15, 10, 2), (15, 19, 10, 2), (19, 23, 12, 2), (23, 27, 11, 1), (15, 15, 9, 1), (27, 27, 11, 1),
                                  (1, 10, 21, 2), (10, 14, 21, 2), (14, 18, 21, 2), (18, 18, 21, 1)):
        assert refnet.conv_rounds_to_bf16(cin, cout, kw, stride, 80), (cin, cout, kw, stride)
