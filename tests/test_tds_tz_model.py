"""The index arithmetic of the block-Toeplitz TDS convolution kernels (conv_tds_tz.hpp forward / backward-data,
conv_tds_tzf.hpp + tds_tzf_reduce_k filter gradient) on the CPU: oracle/tds_tz_model.py walks the kernels' decomposition lane
by lane in float64 (slab addresses, lane-half frame pairing, split tail, B-register order, MFMA ownership, store offsets
and range checks, the partial-image layout and the fold of D's diagonals into dW) with never-written bytes held as NaN,
and is held here to a direct convolution.  The HIP kernels themselves are compared with the oracle under -m gpu
(tests/test_gpu_nn.py::test_tds_conv_streamed_wave_specialised_kernel, test_conv_fwd_bwd, tests/test_gpu_parity_shapes.py)."""
import numpy as np
import pytest

from oracle import tds_tz_model as M


@pytest.mark.parametrize("C,T,B,H,kw,padl,flip,relu,use_add", [
    (10, 30, 1, 16, 21, 10, False, True, False),     # one strip, two rounds, the last one ragged
    (14, 19, 1, 16, 21, 10, False, False, False),
    (18, 14, 1, 16, 21, 10, False, True, False),     # two column tiles, wave pairs
    (10, 7, 2, 32, 9, 0, True, False, True),         # backward-data: flipped transposed weights, addend, short kernel, causal padding
    (18, 27, 1, 16, 21, 20, True, False, True),
    (14, 40, 1, 16, 5, 2, True, False, True),
    (10, 1, 1, 16, 21, 10, False, False, False),     # one frame
])
def test_forward_decomposition(C, T, B, H, kw, padl, flip, relu, use_add):
    rng = np.random.default_rng(C * 100 + T)
    x = rng.normal(size=(B, T, H, C))
    w = rng.normal(size=(kw, C, C))
    bias = None if use_add else rng.normal(size=C)
    add = rng.normal(size=(B, T, H, C)) if use_add else None
    y = M.forward(x, w, bias, kw, padl, flip, relu, add)
    ref = M.direct(x, w, bias, kw, padl, flip, relu, add)
    assert not np.isnan(y).any()          # every output written (exactly once: asserted inside), no read of an unwritten byte
    assert np.abs(y - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("CI,CO,T,B,H,kw,padl,padr", [(10, 14, 41, 1, 16, 21, 10, 10), (14, 18, 37, 2, 16, 21, 9, 10), (10, 14, 60, 1, 32, 21, 10, 9),
                                                      (14, 18, 8, 1, 16, 7, 3, 3)])
def test_strided_layers_decomposition(CI, CO, T, B, H, kw, padl, padr):
    """the sub-sampling layers between the TDS stages (stride 2, CI != CO): forward with a group step of 2 R input frames and
    the Toeplitz index s - 2 r (padded frame pitch: the tail's padding pair must read zeros, not stale LDS); backward-data as one
    launch per phase of the stride (every second tap, every second output frame), with and without the addend"""
    rng = np.random.default_rng(CI * 100 + T)
    To = (T + padl + padr - kw) // 2 + 1
    x = rng.normal(size=(B, T, H, CI))
    w = rng.normal(size=(kw, CI, CO))
    bias = rng.normal(size=CO)
    y = M.forward(x, w, bias, kw, padl, False, True, None, To, stride=2)
    ref = M.direct(x, w, bias, kw, padl, False, True, None, To, stride=2)
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    dy = rng.normal(size=(B, To, H, CO))
    add = rng.normal(size=(B, T, H, CI))
    for a in (None, add):
        dx = M.backward_data_strided(dy, w, T, kw, 2, padl, a)
        rd = M.direct_backward_data(dy, w, T, kw, 2, padl, a)
        assert not np.isnan(dx).any()
        assert np.abs(dx - rd).max() < 1e-12 * max(1.0, np.abs(rd).max())


@pytest.mark.parametrize("C,T,B,H,kw,padl", [(10, 30, 1, 16, 21, 10), (14, 19, 2, 16, 21, 10), (10, 53, 1, 32, 9, 0), (14, 40, 1, 16, 21, 20),
                                             (10, 1, 1, 16, 21, 10)])
def test_filter_gradient_decomposition(C, T, B, H, kw, padl):
    rng = np.random.default_rng(C * 10 + T)
    x = rng.normal(size=(B, T, H, C))
    dy = rng.normal(size=(B, T, H, C))
    dW, db = M.filter_grad(x, dy, kw, padl)
    rW, rb = M.filter_direct(x, dy, kw, padl)
    assert np.abs(dW - rW).max() < 1e-12 * max(1.0, np.abs(rW).max())
    assert np.abs(db - rb).max() < 1e-12 * max(1.0, np.abs(rb).max())


@pytest.mark.parametrize("CI,CO,T,B,H,kw,padl,padr", [(10, 14, 41, 1, 16, 21, 10, 10), (14, 18, 37, 2, 16, 21, 9, 10), (10, 14, 75, 1, 32, 21, 10, 9),
                                                      (14, 18, 30, 1, 16, 7, 3, 3)])
def test_strided_filter_gradient_decomposition(CI, CO, T, B, H, kw, padl, padr):
    """the filter gradient of the stride-2 layers: x groups 2 R input frames apart, D's diagonals j + 2 r"""
    rng = np.random.default_rng(CI * 10 + T)
    To = (T + padl + padr - kw) // 2 + 1
    x = rng.normal(size=(B, T, H, CI))
    dy = rng.normal(size=(B, To, H, CO))
    dW, db = M.filter_grad(x, dy, kw, padl, stride=2)
    rW, rb = M.filter_direct(x, dy, kw, padl, stride=2)
    assert np.abs(dW - rW).max() < 1e-12 * max(1.0, np.abs(rW).max())
    assert np.abs(db - rb).max() < 1e-12 * max(1.0, np.abs(rb).max())
