"""Thin torch-tensor front-ends of the network-operator C ABI (include/w2l_hip.h
section 2).  Used by tests and by the Python side of bench.py; the C++ host
(wav2letter_amd/csrc/host) calls the same kernels directly.  Frame-major
activations: [B][T][H][C]."""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


def gemm(A, B, a_kcontig=True, b_kcontig=False, bias=None, relu=False, splitk=1):
    """C = op(A) op(B): A [M][K] if a_kcontig else [K][M]; B [N][K] if b_kcontig else [K][N]"""
    M, K = (A.shape if a_kcontig else A.shape[::-1])
    N = B.shape[0] if b_kcontig else B.shape[1]
    Cm = torch.empty(M, N, device=A.device, dtype=torch.float32)
    check(_lib.lib().w2l_gemm_f32(M, N, K, _p(A), A.stride(0), int(a_kcontig), _p(B), B.stride(0), int(b_kcontig),
                                  _p(Cm), N, _p(bias), int(relu), splitk, _s()), "gemm")
    return Cm


def linear_forward(x, w, bias=None, relu=False):
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_linear_forward(M, K, N, _p(x), _p(w), _p(bias), _p(y), int(relu), _s()), "linear_forward")
    return y


def weightnorm_forward(v, g):
    """fl::WeightNorm on the internal [K][Nout] weight layout: w = v * g / ||v|| per column; returns (w, norm)"""
    K, N = v.shape
    w = torch.empty_like(v)
    norm = torch.empty(N, device=v.device, dtype=torch.float32)
    check(_lib.lib().w2l_weightnorm_forward(_p(v), _p(g), _p(w), _p(norm), K, N, _s()), "weightnorm_forward")
    return w, norm


def weightnorm_backward(v, g, norm, dw):
    K, N = v.shape
    dv = torch.empty_like(v)
    dg = torch.empty(N, device=v.device, dtype=torch.float32)
    dot = torch.empty(N, device=v.device, dtype=torch.float32)
    check(_lib.lib().w2l_weightnorm_backward(_p(v), _p(g), _p(norm), _p(dw), _p(dv), _p(dg), _p(dot), K, N, _s()), "weightnorm_backward")
    return dv, dg


def colsum(x):
    """out[n] = sum_m x[m][n] (bias gradient): deterministic two-pass sum"""
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_colsum(_p(x), _p(out), M, N, _s()), "colsum")
    return out


def linear_backward(x, w, dy, mask_src=None, mask_scale=1.0):
    M, K = x.shape
    N = w.shape[1]
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    db = torch.empty(N, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    check(L.w2l_linear_backward_data(M, K, N, _p(dy), _p(w), _p(dx), 0, _p(mask_src), mask_scale, _s()), "linear_bwd_data")
    check(L.w2l_linear_backward_weight_bias(M, K, N, _p(x), _p(dy), _p(dw), _p(db), _s()), "linear_bwd_weight_bias")
    return dx, dw, db


def conv_desc(x, w, stride, padl, padr):
    B, T, H, Cin = x.shape
    kw, _, Cout = w.shape
    return ConvDesc(B, T, H, Cin, Cout, kw, stride, padl, padr)


def conv_forward(x, w, bias=None, stride=1, padl=0, padr=0, relu=False):
    """x [B][T][H][Cin], w [kw][Cin][Cout] -> y [B][To][H][Cout]"""
    d = conv_desc(x, w, stride, padl, padr)
    To = _lib.lib().w2l_conv_out_len(d.T, d.kw, stride, padl, padr)
    y = torch.empty(d.B, To, d.H, d.Cout, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_conv_forward(C.byref(d), _p(x), _p(w), _p(bias), _p(y), int(relu), _s()), "conv_forward")
    return y


def conv_backward(x, w, dy, stride=1, padl=0, padr=0, need_dx=True):
    d = conv_desc(x, w, stride, padl, padr)
    L = _lib.lib()
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w)
    db = torch.empty(d.Cout, device=x.device, dtype=torch.float32)
    if need_dx:
        check(L.w2l_conv_backward_data(C.byref(d), _p(dy), _p(w), _p(dx), 0, _s()), "conv_bwd_data")
    check(L.w2l_conv_backward_filter(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _s()), "conv_bwd_filter")
    return dx, dw, db


def _ln_scratch(groups, inner):
    return int(_lib.lib().w2l_layernorm_scratch_doubles(groups, inner))


def residual_layernorm_forward(a, x, gamma_beta, groups, eps=1e-5, p=0.0, seed=0, stream_id=0):
    """returns (y, r, mean_rstd); `a` is dropped in place when p > 0"""
    inner = a.numel() // groups
    r = torch.empty_like(a)
    y = torch.empty_like(a)
    stats = torch.empty(_ln_scratch(groups, inner), device=a.device, dtype=torch.float64)
    mr = torch.empty(2 * groups, device=a.device, dtype=torch.float32)
    check(_lib.lib().w2l_residual_layernorm_forward(groups, inner, _p(a), _p(x), _p(r), _p(y), _p(gamma_beta), eps,
                                                    p, seed, stream_id, _p(stats), _p(mr), _s()), "res_ln_fwd")
    return y, r, mr


def layernorm_backward(r, dy, gamma_beta, mean_rstd, groups, mask_src=None, mask_scale=1.0):
    inner = r.numel() // groups
    dr = torch.empty_like(r)
    dgb = torch.empty(2, device=r.device, dtype=torch.float32)
    dmask = torch.empty_like(r) if mask_src is not None else None
    sums = torch.empty(_ln_scratch(groups, inner), device=r.device, dtype=torch.float64)
    check(_lib.lib().w2l_layernorm_backward(groups, inner, _p(r), _p(dy), _p(gamma_beta), _p(mean_rstd), _p(dr),
                                            _p(dgb), _p(mask_src), _p(dmask), mask_scale, _p(sums), _s()), "ln_bwd")
    return dr, dgb, dmask


def linear_forward_dropout(x, w, bias, relu, p, seed, stream_id):
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_linear_forward_dropout(M, K, N, _p(x), _p(w), _p(bias), _p(y), int(relu), p, seed, stream_id, _s()), "linear_forward_dropout")
    return y


def linear_forward_dropout_add(x, w, bias, add, relu, p, seed, stream_id):
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_linear_forward_dropout_add(M, K, N, _p(x), _p(w), _p(bias), _p(add), _p(y), int(relu), p, seed, stream_id, _s()),
          "linear_forward_dropout_add")
    return y


def linear_backward_data_add(dy, w, add):
    M, N = dy.shape
    K = w.shape[0]
    dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
    check(_lib.lib().w2l_linear_backward_data_add(M, K, N, _p(dy), _p(w), _p(add), _p(dx), _s()), "linear_bwd_data_add")
    return dx


def dropout_copy(x, p, seed, stream_id):
    y = torch.empty_like(x)
    check(_lib.lib().w2l_dropout_copy(_p(y), _p(x), x.numel(), p, seed, stream_id, _s()),
          "dropout_copy")
    return y


def dropout_(x, p, seed, stream_id):
    check(_lib.lib().w2l_dropout_inplace(_p(x), x.numel(), p, seed, stream_id, _s()), "dropout")
    return x


def transpose(x):
    """[G][R][C] -> [G][C][R]"""
    G, R, Cc = x.shape
    out = torch.empty(G, Cc, R, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_transpose(_p(x), _p(out), G, R, Cc, _s()), "transpose")
    return out


def glu_forward(x):
    M, two = x.shape
    y = torch.empty(M, two // 2, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_glu_forward(_p(x), _p(y), M, two // 2, _s()), "glu_fwd")
    return y


def glu_backward(x, dy):
    M, two = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().w2l_glu_backward(_p(x), _p(dy), _p(dx), M, two // 2, _s()), "glu_bwd")
    return dx


def sgd_step_(p, g, v, lr, momentum, grad_scale=1.0, max_grad_norm=0.0):
    L = _lib.lib()
    ss = torch.zeros(1, device=p.device, dtype=torch.float64)
    if max_grad_norm > 0:
        check(L.w2l_sumsq(_p(g), g.numel(), _p(ss), 1, _s()), "sumsq")
    check(L.w2l_sgd_step(_p(p), _p(g), _p(v), p.numel(), lr, momentum, grad_scale, max_grad_norm, _p(ss), _s()), "sgd")


def bf16_convert(x, rows_image=True, transposed_image=False):
    """w2l_bf16_convert: x [rows][cols] fp32 -> (rowMajor [rows][colsP], transposed [cols][rowsP]) bf16 images, zero-padded to
    the next multiple of 64 (None for an image that was not asked for).  The images are torch.bfloat16 tensors."""
    x = x.contiguous()
    rows, cols = x.shape
    colsP, rowsP = (cols + 63) // 64 * 64, (rows + 63) // 64 * 64
    rm = torch.empty(rows, colsP, dtype=torch.bfloat16, device=x.device) if rows_image else None
    tr = torch.empty(cols, rowsP, dtype=torch.bfloat16, device=x.device) if transposed_image else None
    _lib.check(_lib.lib().w2l_bf16_convert(_p(x), rows, cols, cols, _p(rm) if rm is not None else None, colsP,
                                           _p(tr) if tr is not None else None, rowsP, _s()), "bf16_convert")
    return rm, tr


def bf16_convert_dropout(x, p, seed, stream_id, rows_image=True, transposed_image=True):
    """w2l_bf16_convert_dropout: the bf16 images of dropout(x) (mask of dropout_copy(p, seed, stream_id)) in one pass over x"""
    x = x.contiguous()
    rows, cols = x.shape
    colsP, rowsP = (cols + 63) // 64 * 64, (rows + 63) // 64 * 64
    rm = torch.empty(rows, colsP, dtype=torch.bfloat16, device=x.device) if rows_image else None
    tr = torch.empty(cols, rowsP, dtype=torch.bfloat16, device=x.device) if transposed_image else None
    _lib.check(_lib.lib().w2l_bf16_convert_dropout(_p(x), rows, cols, cols, _p(rm) if rm is not None else None, colsP,
                                                   _p(tr) if tr is not None else None, rowsP, p, seed, stream_id, _s()), "bf16_convert_dropout")
    return rm, tr


def gemm_bf16(A, B, K, bias=None, relu=False, mask=None, mask_scale=1.0, addend=None, out=None, accumulate=False,
              drop_p=0.0, drop_seed=0, drop_stream=0):
    """w2l_gemm_bf16: C[M][N] fp32 = A[M][>=K] . B[N][>=K]^T on bf16 images (rows zero from column K to the next multiple of 64)"""
    import ctypes as C
    M, N = A.shape[0], B.shape[0]
    c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    e = _lib.GemmEpilogue()
    e.mask = _p(mask) if mask is not None else None
    e.maskScale = mask_scale
    e.addend = _p(addend) if addend is not None else None
    e.accumulate = int(accumulate)
    e.dropP = drop_p
    e.dropSeed, e.dropStream = drop_seed, drop_stream
    _lib.check(_lib.lib().w2l_gemm_bf16(M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(c), c.stride(0),
                                        _p(bias) if bias is not None else None, int(relu), C.byref(e), _s()), "gemm_bf16")
    return c


def gemm_bf16_images(A, B, K, bias=None, relu=False, keep_f32=False, rows_image=True, transposed_image=True, mask_image=None,
                     mask_scale=1.0, addend=None, drop_p=0.0, drop_seed=0, drop_stream=0):
    """w2l_gemm_bf16_images: the product's result as bf16 images (zero-padded to multiples of 64 here, the caller's job in the ABI)
    instead of / beside the fp32 C; mask_image: a bf16 row-major image as the mask operand.  Returns (C or None, rows, transposed)."""
    import ctypes as C
    M, N = A.shape[0], B.shape[0]
    colsP, rowsP = (N + 63) // 64 * 64, (M + 63) // 64 * 64
    c = torch.empty(M, N, dtype=torch.float32, device=A.device) if keep_f32 else None
    rm = torch.zeros(M, colsP, dtype=torch.bfloat16, device=A.device) if rows_image else None
    tr = torch.zeros(N, rowsP, dtype=torch.bfloat16, device=A.device) if transposed_image else None
    e = _lib.GemmEpilogue()
    e.mask = None; e.maskScale = 1.0
    e.addend = _p(addend) if addend is not None else None
    e.accumulate = 0
    e.dropP = drop_p
    e.dropSeed, e.dropStream = drop_seed, drop_stream
    k = _lib.Bf16ImageSink()
    k.rowMajor = _p(rm) if rm is not None else None; k.ldRows = colsP
    k.transposed = _p(tr) if tr is not None else None; k.ldTrans = rowsP
    _lib.check(_lib.lib().w2l_gemm_bf16_images(M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(c) if c is not None else None, N,
                                               _p(bias) if bias is not None else None, int(relu), C.byref(e),
                                               C.byref(k) if (rm is not None or tr is not None) else None,
                                               _p(mask_image) if mask_image is not None else None,
                                               mask_image.stride(0) if mask_image is not None else 0, mask_scale, _s()), "gemm_bf16_images")
    return c, rm, tr


def gemm_bf16_ex(A, B, M, N, K, a_kmajor=False, b_kmajor=False, bias=None, relu=False):
    """w2l_gemm_bf16_ex: C[M][N] fp32 = op(A) op(B)^T with k-MAJOR operands read in place: a_kmajor -> A is [K][>= M],
    b_kmajor -> B is [K][>= N] (row strides multiples of 8 elements); otherwise as gemm_bf16"""
    c = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().w2l_gemm_bf16_ex(M, N, K, _p(A), A.stride(0), int(a_kmajor), _p(B), B.stride(0), int(b_kmajor), _p(c), c.stride(0),
                                           _p(bias) if bias is not None else None, int(relu), None, _s()), "gemm_bf16_ex")
    return c


def gemm_bf16_grouped(As, Bs, K, biases=None):
    """w2l_gemm_bf16_grouped: [C_g = A_g . B_g^T (+ bias_g)] for up to 4 problems of one shape in one launch"""
    import ctypes as C
    n = len(As)
    M, N = As[0].shape[0], Bs[0].shape[0]
    outs = [torch.empty(M, N, dtype=torch.float32, device=As[0].device) for _ in range(n)]
    arr = lambda ts: (C.c_void_p * n)(*[(_p(t) if t is not None else None) for t in ts])
    _lib.check(_lib.lib().w2l_gemm_bf16_grouped(n, M, N, K, arr(As), As[0].stride(0), arr(Bs), Bs[0].stride(0), arr(outs), N,
                                                arr(biases) if biases is not None else None, _s()), "gemm_bf16_grouped")
    return outs


def tds_conv_bf16(x, w, bias, padl, padr, relu=False, stride=1):
    """the TDS / sub-sampling convolution on bf16-rounded operands (w2l_tds_conv_bf16_*): x [B][T][H][Cin], w [kw][Cin][Cout] ->
    y [B][To][H][Cout]; None when the geometry has no bf16 kernel"""
    import ctypes as C
    d = conv_desc(x, w, stride, padl, padr)
    n = _lib.lib().w2l_tds_conv_bf16_image_elems(C.byref(d))
    if not n:
        return None
    imgs = torch.empty(2, n, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().w2l_tds_conv_bf16_prepare(C.byref(d), _p(w), _p(imgs[0]), _p(imgs[1]), _s()), "tds_conv_bf16_prepare")
    To = _lib.lib().w2l_conv_out_len(d.T, d.kw, stride, padl, padr)
    y = torch.empty(d.B, To, d.H, d.Cout, device=x.device, dtype=torch.float32)
    check(_lib.lib().w2l_tds_conv_bf16_forward(C.byref(d), _p(x), _p(imgs[0]), _p(bias), _p(y), int(relu), _s()), "tds_conv_bf16_forward")
    return y, imgs, d


def tds_conv_bf16_backward(x, dy, imgs, d, add=None, with_bias=False):
    """(dx, dw[, db]) of the same convolution: dx = add + conv^T(dy), dw = x (*) dy, both on bf16-rounded operands; with_bias: the
    bias gradient (column sums of the rounded dy) from the same launch"""
    import ctypes as C
    dx = torch.empty_like(x)
    check(_lib.lib().w2l_tds_conv_bf16_backward_data(C.byref(d), _p(dy), _p(imgs[1]), _p(add), _p(dx), _s()), "tds_conv_bf16_backward_data")
    dw = torch.empty(d.kw, d.Cin, d.Cout, dtype=torch.float32, device=x.device)
    if with_bias:
        db = torch.empty(d.Cout, dtype=torch.float32, device=x.device)
        check(_lib.lib().w2l_tds_conv_bf16_backward_filter_bias(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _s()), "tds_conv_bf16_backward_filter_bias")
        return dx, dw, db
    check(_lib.lib().w2l_tds_conv_bf16_backward_filter(C.byref(d), _p(x), _p(dy), _p(dw), _s()), "tds_conv_bf16_backward_filter")
    return dx, dw
