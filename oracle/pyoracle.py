"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product package
(wav2letter_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SCALE_NONE, SCALE_INPUT_SZ, SCALE_INPUT_SZ_SQRT, SCALE_TARGET_SZ, SCALE_TARGET_SZ_SQRT = range(5)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("criterion_oracle.c", "nn_oracle.c")]
    if force or not os.path.exists(so) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        if all(os.path.exists(s) for s in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-B", "-s"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.w2l_oracle_fcc_workspace_size.restype = C.c_size_t
        _LIB.w2l_oracle_fac_workspace_size.restype = C.c_size_t
        _LIB.w2l_oracle_ctc_workspace_size.restype = C.c_size_t
        _LIB.w2l_oracle_dropout_threshold.restype = C.c_uint32
        _LIB.w2l_oracle_dropout_threshold.argtypes = [C.c_double]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------- criteria
def batch_target_size(target, max_size):
    target = _i32(target)
    B, L = target.shape
    out = np.zeros(B, np.int32)
    lib().w2l_oracle_batch_target_size(B, L, int(max_size), _p(target), _p(out))
    return out


def batch_ctc_target_size(target, T):
    target = _i32(target)
    B, L = target.shape
    out = np.zeros(B, np.int32)
    lib().w2l_oracle_batch_ctc_target_size(B, L, int(T), _p(target), _p(out))
    return out


class FCC:
    """FullConnectionCriterion oracle: forward() then backward()."""

    def __init__(self, x, trans, target_size, scale_mode=SCALE_NONE):
        self.x = _f32(x)
        self.trans = _f32(trans)
        self.B, self.T, self.N = self.x.shape
        self.ts = _i32(target_size)
        self.mode = scale_mode
        self.ws = np.zeros(lib().w2l_oracle_fcc_workspace_size(self.B, self.T, self.N), np.uint8)

    def forward(self):
        loss = np.zeros(self.B, np.float64)
        lib().w2l_oracle_fcc_forward(self.B, self.T, self.N, self.mode, _p(self.x), _p(self.ts),
                                     _p(self.trans), _p(loss), _p(self.ws))
        return loss

    def backward(self, grad=None):
        grad = _f64(np.ones(self.B) if grad is None else grad)
        dx = np.zeros((self.B, self.T, self.N), np.float64)
        dt = np.zeros((self.N, self.N), np.float64)
        lib().w2l_oracle_fcc_backward(self.B, self.T, self.N, _p(self.trans), _p(grad), _p(dx),
                                      _p(dt), _p(self.ws))
        return dx, dt


class FAC:
    """ForceAlignmentCriterion oracle."""

    def __init__(self, x, trans, target, target_size=None, scale_mode=SCALE_NONE):
        self.x = _f32(x)
        self.trans = _f32(trans)
        self.target = _i32(target)
        self.B, self.T, self.N = self.x.shape
        self.L = self.target.shape[1]
        self.ts = _i32(batch_target_size(self.target, self.T) if target_size is None else target_size)
        self.mode = scale_mode
        self.ws = np.zeros(lib().w2l_oracle_fac_workspace_size(self.B, self.T, self.N, self.L), np.uint8)

    def forward(self):
        loss = np.zeros(self.B, np.float64)
        lib().w2l_oracle_fac_forward(self.B, self.T, self.N, self.L, self.mode, _p(self.x),
                                     _p(self.target), _p(self.ts), _p(self.trans), _p(loss), _p(self.ws))
        return loss

    def backward(self, grad=None):
        grad = _f64(np.ones(self.B) if grad is None else grad)
        dx = np.zeros((self.B, self.T, self.N), np.float64)
        dt = np.zeros((self.N, self.N), np.float64)
        lib().w2l_oracle_fac_backward(self.B, self.T, self.N, self.L, _p(self.target), _p(self.ts),
                                      _p(self.trans), _p(grad), _p(dx), _p(dt), _p(self.ws))
        return dx, dt

    def viterbi(self):
        path = np.zeros((self.B, self.T), np.int32)
        lib().w2l_oracle_fac_viterbi(self.B, self.T, self.N, self.L, _p(self.x), _p(self.target),
                                     _p(self.ts), _p(self.trans), _p(path))
        return path


def asg(x, trans, target, scale_mode=SCALE_NONE, grad=None):
    """ASG = FCC - FAC sharing the transitions. Returns loss, dx, dtrans (fp64)."""
    fac = FAC(x, trans, target, scale_mode=scale_mode)
    fcc = FCC(x, trans, fac.ts, scale_mode)
    loss = fcc.forward() - fac.forward()
    dx1, dt1 = fcc.backward(grad)
    dx2, dt2 = fac.backward(grad)
    return loss, dx1 - dx2, dt1 - dt2


def linear_target(target, T):
    """Flashlight getLinearTarget (CriterionUtils, [UNVENDORED]; used by LinearSegmentationCriterion::forward, the
    reference's LinSegCriterion of recipes/slimIPL/src/Train.cpp:592): newTarget[b][t] = target[b][t * L_b / T] with
    L_b = leading non-negative entries; a row with L_b == 0 or L_b > T is filled with -1.  Parity unpinned in
    /root/reference (no LinSeg test there)."""
    target = np.asarray(target, np.int32)
    B, L = target.shape
    out = np.full((B, T), -1, np.int32)
    for b in range(B):
        neg = np.nonzero(target[b] < 0)[0]
        tn = int(neg[0]) if len(neg) else L
        if tn == 0 or tn > T:
            continue
        out[b] = target[b][(np.arange(T, dtype=np.int64) * tn) // T]
    return out


def linseg(x, trans, target, scale_mode=SCALE_NONE, grad=None):
    """LinSegCriterion = ASG on the linearly stretched target (rows that cannot be stretched: FAC part 0, FCC with
    target size 0, as the HIP FAC / FCC do for an empty target)"""
    x = _f32(x)
    lin = linear_target(target, x.shape[1])
    return asg(x, trans, lin, scale_mode, grad)


def mfsc_filterbank(num_filters, nfft, fs, low_hz=0.0, high_hz=None):
    """Flashlight TriFilterbank (MEL), [UNVENDORED] -- recalled: numFilters + 2 points equally spaced on the mel scale
    2595 log10(1 + f/700), converted back to Hz and then to FFT-bin units; filter j is the triangle
    max(min((k - f_j)/(f_j+1 - f_j), (f_j+2 - k)/(f_j+2 - f_j+1)), 0) over the bins k = 0..nfft/2.  Returns H [nfft/2+1][num_filters]."""
    high_hz = fs / 2.0 if high_hz is None else high_hz
    nb = nfft // 2 + 1
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    imel = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    pts = imel(np.linspace(mel(low_hz), mel(high_hz), num_filters + 2)) * (nb - 1) * 2.0 / fs
    k = np.arange(nb, dtype=np.float64)[:, None]
    hi = (k - pts[None, :-2]) / (pts[None, 1:-1] - pts[None, :-2])
    lo = (pts[None, 2:] - k) / (pts[None, 2:] - pts[None, 1:-1])
    return np.maximum(np.minimum(hi, lo), 0.0)


def mfsc(audio, num_filters=80, fs=16000, frame_ms=25, stride_ms=10, preem=0.97, use_power=False, mel_floor=1.0):
    """fl::lib::audio::Mfsc as LogMelFeature configures it (LogMelFeature.cpp:78-95; arithmetic [UNVENDORED], recalled:
    PowerSpectrum = frames of round(fs*25ms) samples every round(fs*10ms), no dither, no mean removal, in-frame
    pre-emphasis x[i] -= 0.97 x[i-1] (x[0] *= 0.03), Hamming window 0.54 - 0.46 cos(2 pi i/(N-1)), |FFT| at the next
    power of two (usePower = false: magnitude), TriFilterbank, max(., melFloor = 1), natural log).  One utterance
    [n_samples] -> [T][num_filters], frame by frame in fp64.  PARITY UNPINNED: the reference tree only tests that the
    features do not depend on how the audio is chunked (LogMelFeatureTest.cpp:25-66)."""
    x = np.asarray(audio, np.float64)
    N = int(round(1e-3 * frame_ms * fs)); S = int(round(1e-3 * stride_ms * fs))
    nfft = 1 << (N - 1).bit_length()
    T = 0 if len(x) < N else 1 + (len(x) - N) // S
    H = mfsc_filterbank(num_filters, nfft, fs)
    win = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(N) / (N - 1))
    out = np.zeros((T, num_filters))
    for t in range(T):
        f = x[t * S:t * S + N].copy()
        f[1:] -= preem * f[:-1]
        f[0] *= 1.0 - preem
        spec = np.abs(np.fft.rfft(f * win, nfft))
        if use_power:
            spec = spec ** 2
        out[t] = np.log(np.maximum(spec @ H, mel_floor))
    return out


def viterbi(x, trans):
    x = _f32(x)
    trans = _f32(trans)
    B, T, N = x.shape
    path = np.zeros((B, T), np.int32)
    lib().w2l_oracle_viterbi_compute(B, T, N, _p(x), _p(trans), _p(path))
    return path


class CTC:
    def __init__(self, x, target, target_size=None, scale_mode=SCALE_NONE):
        self.x = _f32(x)
        self.target = _i32(target)
        self.B, self.T, self.N = self.x.shape
        self.L = self.target.shape[1]
        self.ts = _i32(batch_ctc_target_size(self.target, self.T) if target_size is None else target_size)
        self.mode = scale_mode
        self.ws = np.zeros(lib().w2l_oracle_ctc_workspace_size(self.B, self.T, self.N, self.L), np.uint8)

    def forward(self):
        loss = np.zeros(self.B, np.float64)
        lib().w2l_oracle_ctc_forward(self.B, self.T, self.N, self.L, self.mode, _p(self.x),
                                     _p(self.target), _p(self.ts), _p(loss), _p(self.ws))
        return loss

    def backward(self, grad=None):
        grad = _f64(np.ones(self.B) if grad is None else grad)
        dx = np.zeros((self.B, self.T, self.N), np.float64)
        lib().w2l_oracle_ctc_backward(self.B, self.T, self.N, self.L, _p(self.x), _p(self.target),
                                      _p(self.ts), _p(grad), _p(dx), _p(self.ws))
        return dx


def ctc_viterbi(x):
    x = _f32(x)
    B, T, N = x.shape
    path = np.zeros((B, T), np.int32)
    lib().w2l_oracle_ctc_viterbi(B, T, N, _p(x), _p(path))
    return path


# ----------------------------------------------------------------- network ops
def conv_out_len(T, kw, stride, padl, padr, dil=1):
    return lib().w2l_oracle_conv_out_len(T, kw, stride, padl, padr, dil)


def same_pad(T, kw, stride, dil=1):
    return lib().w2l_oracle_same_pad(T, kw, stride, dil)


def conv_fwd(x, w, bias, stride=1, padl=0, padr=0, dil=1):
    """x [B][Cin][H][T], w [Cout][Cin][kw] -> y [B][Cout][H][To]"""
    x = _f32(x); w = _f32(w)
    B, Cin, H, T = x.shape
    Cout, _, kw = w.shape
    To = conv_out_len(T, kw, stride, padl, padr, dil)
    y = np.zeros((B, Cout, H, To), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().w2l_oracle_conv_fwd(_p(x), _p(w), _p(b), _p(y), B, Cin, Cout, H, T, kw, stride, padl, padr, dil)
    return y


def conv_bwd(x, w, dy, stride=1, padl=0, padr=0, dil=1, need_dx=True):
    x = _f32(x); w = _f32(w); dy = _f32(dy)
    B, Cin, H, T = x.shape
    Cout, _, kw = w.shape
    dx = np.zeros_like(x) if need_dx else None
    dw = np.zeros_like(w)
    db = np.zeros(Cout, np.float32)
    if need_dx:
        lib().w2l_oracle_conv_bwd_data(_p(dy), _p(w), _p(dx), B, Cin, Cout, H, T, kw, stride, padl, padr, dil)
    lib().w2l_oracle_conv_bwd_filter(_p(x), _p(dy), _p(dw), _p(db), B, Cin, Cout, H, T, kw, stride,
                                     padl, padr, dil)
    return dx, dw, db


def linear_fwd(x, w, bias):
    """x [M][in], w [in][out] (Flashlight dims (out,in)), bias [out]"""
    x = _f32(x); w = _f32(w)
    M, K = x.shape
    N = w.shape[1]
    y = np.zeros((M, N), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().w2l_oracle_linear_fwd(_p(x), _p(w), _p(b), _p(y), M, K, N)
    return y


def linear_bwd(x, w, dy):
    x = _f32(x); w = _f32(w); dy = _f32(dy)
    M, K = x.shape
    N = w.shape[1]
    dx = np.zeros_like(x); dw = np.zeros_like(w); db = np.zeros(N, np.float32)
    lib().w2l_oracle_linear_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), M, K, N)
    return dx, dw, db


def layernorm_fwd(x, groups, gamma=1.0, beta=0.0, eps=1e-5, streaming=False):
    x = _f32(x)
    inner = x.size // groups
    y = np.zeros_like(x)
    lib().w2l_oracle_layernorm_fwd(_p(x), _p(y), None, None, groups, C.c_size_t(inner),
                                   C.c_float(gamma), C.c_float(beta), C.c_float(eps), int(streaming))
    return y


def layernorm_bwd(x, dy, groups, gamma=1.0, eps=1e-5):
    x = _f32(x); dy = _f32(dy)
    inner = x.size // groups
    dx = np.zeros_like(x)
    dg = C.c_double(0); db = C.c_double(0)
    lib().w2l_oracle_layernorm_bwd(_p(x), _p(dy), _p(dx), C.byref(dg), C.byref(db), groups,
                                   C.c_size_t(inner), C.c_float(gamma), C.c_float(eps))
    return dx, dg.value, db.value


def glu_fwd(x, outer, half, inner):
    x = _f32(x)
    y = np.zeros(outer * half * inner, np.float32)
    lib().w2l_oracle_glu_fwd(_p(x), _p(y), C.c_size_t(outer), C.c_size_t(half), C.c_size_t(inner))
    return y


def glu_bwd(x, dy, outer, half, inner):
    x = _f32(x); dy = _f32(dy)
    dx = np.zeros_like(x)
    lib().w2l_oracle_glu_bwd(_p(x), _p(dy), _p(dx), C.c_size_t(outer), C.c_size_t(half), C.c_size_t(inner))
    return dx


def weightnorm_fwd(v, g, outer, nout, inner):
    v = _f32(v); g = _f32(g)
    w = np.zeros_like(v)
    lib().w2l_oracle_weightnorm_fwd(_p(v), _p(g), _p(w), None, C.c_size_t(outer), C.c_size_t(nout),
                                    C.c_size_t(inner))
    return w


def weightnorm_bwd(v, g, dw, outer, nout, inner):
    v = _f32(v); g = _f32(g); dw = _f32(dw)
    dv = np.zeros_like(v); dg = np.zeros_like(g)
    lib().w2l_oracle_weightnorm_bwd(_p(v), _p(g), _p(dw), _p(dv), _p(dg), C.c_size_t(outer),
                                    C.c_size_t(nout), C.c_size_t(inner))
    return dv, dg


def dropout(x, p, seed, stream):
    x = _f32(x)
    y = np.zeros_like(x)
    lib().w2l_oracle_dropout(_p(x), _p(y), C.c_size_t(x.size), C.c_double(p), C.c_uint32(seed),
                             C.c_uint32(stream))
    return y


def specaugment(x, f_mask_f, n_f_mask, t_mask_t, t_mask_p, n_t_mask, seed):
    """fl::SpecAugment on frame-major x [B][T][F] (copy); returns (y, masks) with masks[4][8] = inclusive
    f0, f1, t0, t1 per mask (f1 < f0 / t1 < t0: unused slot)"""
    y = _f32(x).copy()
    B, T, F = y.shape
    masks = np.zeros((4, 8), np.int32)
    st = lib().w2l_oracle_specaugment(_p(y), B, T, F, int(f_mask_f), int(n_f_mask), int(t_mask_t),
                                      C.c_float(t_mask_p), int(n_t_mask), C.c_uint32(seed), _p(masks))
    if st:
        raise ValueError("specaugment: F < fMaskF or more than 8 masks")
    return y, masks


def streaming_conv1d(x, w, bias, T, groups, cin_g, cout_g, kw, stride, padl, padr):
    x = _f32(x); w = _f32(w); bias = _f32(bias)
    To = (T + padl + padr - kw) // stride + 1
    y = np.zeros(To * groups * cout_g, np.float32)
    lib().w2l_oracle_streaming_conv1d(_p(x), _p(w), _p(bias), _p(y), T, groups, cin_g, cout_g, kw,
                                      stride, padl, padr)
    return y


def num_threads():
    return lib().w2l_oracle_num_threads()


def set_num_threads(n):
    lib().w2l_oracle_set_num_threads(int(n))
