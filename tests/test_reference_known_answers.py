"""The known answers and asserted properties of the reference's OTHER module tests -- LinearTest.cpp, ReluTest.cpp, LayerNormTest.cpp,
ResidualTest.cpp under recipes/streaming_convnets/inference/inference/module/test/ (Conv1dTest / TDSBlockTest have their own fixtures)
-- through the oracle on the CPU and through the HIP operators on the device.  The numbers are the reference's
(tests/golden/reference_known_answers.json, extracted by tests/golden/make_golden.py); where the reference test draws random data and
asserts a property (LayerNorm: rows drawn from N(mean_i, std_i) come out as alpha z + beta to 1e-1; Residual: ones in, conv(ones) + 1
out), the same property is asserted on data drawn the same way, at the reference's tolerance."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _linear_cases():
    for c in K["linear"]:
        x = np.array(c["input"], np.float32).reshape(-1, c["nIn"])            # frames x nIn (IOBuffer order: frame-major)
        # createLinear(nIn, nOut, weights, bias): out[t][o] = sum_i w[o][i] x[t][i] + b[o] -- weights are [nOut][nIn] there
        # (Linear.h); the library's (and Flashlight's column-major (out, in)) memory order is [in][out]
        w = np.array(c["weights"], np.float32).reshape(c["nOut"], c["nIn"]).T
        yield x, np.ascontiguousarray(w), np.array(c["bias"], np.float32), c["expected"]


def _ln_data(T, F, seed):
    rng = np.random.default_rng(seed)
    mean, std = rng.random(T).astype(np.float32), rng.random(T).astype(np.float32)   # randVec: uniform [0, 1)
    std = np.maximum(std, 1e-2)                       # (a row of std < 1e-2 says nothing at the reference's absolute tolerance either)
    x = (mean[:, None] + std[:, None] * rng.standard_normal((T, F))).astype(np.float32)
    return x, mean, std


def _residual_data(seed):
    r = K["residual_property"]
    T, G, Cc, kw = r["T"], r["groups"], r["channels"] // r["groups"], r["kernelSize"]
    rng = np.random.default_rng(seed)
    w = rng.random(Cc * kw * Cc).astype(np.float32)    # [co][k][ci] of one group, shared by the groups (Conv1dFbGemm.cpp)
    b = rng.random(Cc).astype(np.float32)
    return r, T, G, Cc, kw, w, b, np.ones(T * G * Cc, np.float32)


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_linear_known_answers_oracle(oracle):
    for x, w, b, expected in _linear_cases():
        y = oracle.linear_fwd(x, w, b).reshape(-1)
        assert len(y) == len(expected)
        for got, (want, tol) in zip(y, expected):
            assert abs(got - want) < tol


def test_relu_known_answers_oracle(oracle):
    x = np.array(K["relu"]["input"], np.float32)[None]
    y = np.maximum(oracle.linear_fwd(x, np.eye(6, dtype=np.float32), None), 0).reshape(-1)      # Relu on the oracle's product with I
    for got, (want, tol) in zip(y, K["relu"]["expected"]):
        assert abs(got - want) < tol


def test_layernorm_property_oracle(oracle):
    p = K["layernorm_property"]
    x, mean, std = _ln_data(p["T"], p["F"], 5)
    for streaming in (True, False):      # the streaming form the reference test runs, and the train-time form (eps inside the sqrt)
        y = oracle.layernorm_fwd(x, p["T"], gamma=p["alpha"], beta=p["beta"], streaming=streaming).reshape(p["T"], p["F"])
        want = p["alpha"] * (x - mean[:, None]) / std[:, None] + p["beta"]
        assert np.abs(y - want).max() < p["tol"]


def test_residual_property_oracle(oracle):
    r, T, G, Cc, kw, w, b, ones = _residual_data(9)
    y = oracle.streaming_conv1d(ones, w, b, T, G, Cc, Cc, kw, r["stride"], r["leftPadding"], r["rightPadding"])
    res = y + ones                                    # Residual.cpp:92-109: module output + its input
    assert np.abs((y + 1.0) - res).max() < r["tol"] and np.isfinite(y).all() and np.abs(y).max() > 0.1


# ------------------------------------------------------------------------------------------------ HIP operators
@pytest.mark.gpu
def test_linear_known_answers_on_device():
    from wav2letter_amd import ops
    for x, w, b, expected in _linear_cases():
        y = ops.linear_forward(dev(x), dev(w), dev(b)).cpu().numpy().reshape(-1)
        assert len(y) == len(expected)
        for got, (want, tol) in zip(y, expected):
            assert abs(got - want) < tol


@pytest.mark.gpu
def test_relu_known_answers_on_device():
    """the reference's Relu vector through the ReLU epilogue of the HIP product against the identity (the library has no stand-alone
    ReLU: it is always the epilogue of the product in front of it)"""
    from wav2letter_amd import ops
    x = np.array(K["relu"]["input"], np.float32)[None]
    y = ops.linear_forward(dev(x), dev(np.eye(6, dtype=np.float32)), dev(np.zeros(6, np.float32)), relu=True).cpu().numpy().reshape(-1)
    for got, (want, tol) in zip(y, K["relu"]["expected"]):
        assert abs(got - want) < tol


@pytest.mark.gpu
def test_layernorm_property_on_device():
    from wav2letter_amd import ops
    p = K["layernorm_property"]
    x, mean, std = _ln_data(p["T"], p["F"], 5)
    gb = dev(np.array([p["alpha"], p["beta"]], np.float32))
    y, _, _ = ops.residual_layernorm_forward(dev(x), None, gb, p["T"], eps=0.0)           # streaming form: no epsilon
    want = p["alpha"] * (x - mean[:, None]) / std[:, None] + p["beta"]
    assert np.abs(y.cpu().numpy().reshape(p["T"], p["F"]) - want).max() < p["tol"]


@pytest.mark.gpu
def test_residual_property_on_device():
    """Residual(conv)(ones) = conv(ones) + 1: the HIP convolution, and the residual sum the LayerNorm kernel forms (its `r` output)"""
    from wav2letter_amd import ops
    r, T, G, Cc, kw, w, b, ones = _residual_data(9)
    xd = dev(ones.reshape(1, T, G, Cc))
    wc = dev(np.ascontiguousarray(w.reshape(Cc, kw, Cc).transpose(1, 2, 0)))              # [co][k][ci] -> [k][ci][co]
    a = ops.conv_forward(xd, wc, dev(b), r["stride"], r["leftPadding"], r["rightPadding"])
    no_res = a.cpu().numpy().reshape(-1)
    _, res, _ = ops.residual_layernorm_forward(a, xd, dev(np.array([1.0, 0.0], np.float32)), T, eps=0.0)
    assert np.abs((no_res + 1.0) - res.cpu().numpy().reshape(-1)).max() < r["tol"]
    assert np.abs(no_res).max() > 0.1
