"""ctypes binding of libw2l_hip.so (the C ABI declared in include/w2l_hip.h).

The library is built in-tree by `__graft_entry__.build()` /
`make -C wav2letter_amd/csrc`.  There is NO fallback: if the shared object is
missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("W2L_HIP_SO") or os.path.join(_HERE, "libw2l_hip.so")  # W2L_HIP_SO: A/B builds of the same ABI
_lib = None

W2L_OK, W2L_EINVAL, W2L_EHIP, W2L_EUNSUPPORTED = 0, 1, 2, 3
_ERR = {1: "W2L_EINVAL (bad shape / null pointer)", 2: "W2L_EHIP (HIP runtime error)",
        3: "W2L_EUNSUPPORTED (shape outside this build)"}


class W2LError(RuntimeError):
    pass


class W2LInvalidArgument(W2LError, ValueError):
    """mirrors the std::invalid_argument Flashlight's criteria throw"""


def _load(path):
    if not os.path.exists(path):
        raise W2LError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C wav2letter_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(path)
    L.w2l_version.restype = C.c_char_p
    for name in dir(_SIGS):
        if name.startswith("w2l_"):
            fn = getattr(L, name)
            restype, argtypes = getattr(_SIGS, name)
            fn.restype = restype
            fn.argtypes = argtypes
    return L


def lib():
    global _lib
    if _lib is None:
        # W2L_LIB_PATH: load another build of the same library (A/B runs of two kernel generations, tools/ only)
        _lib = _load(os.environ.get("W2L_LIB_PATH") or SO_PATH)
    return _lib


PROBE_SO_PATH = os.path.join(_HERE, "libw2l_hip_probe.so")
_probe = None


class use_probe:
    """context manager: route every call of this package through libw2l_hip_probe.so (built with -DW2L_PROBE: the only
    build that honours the W2L_* kernel-variant switches and carries the ablation / experimental kernels).  For tools/
    and the kernel-variant tests; the product library ignores those environment variables."""

    def __enter__(self):
        global _lib, _probe
        if _probe is None:
            _probe = _load(PROBE_SO_PATH)
        self._saved = lib()
        _lib = _probe
        return _probe

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


_p, _i, _sz, _f, _u32, _u64, _d = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_uint32, C.c_uint64, C.c_double


class _SIGS:
    w2l_last_hip_error = (_i, [])
    w2l_flac_info = (_i, [_p, _sz, _p, _p, _p, _p])
    w2l_flac_decode = (_i, [_p, _sz, _p, _u64, _p, _p])
    w2l_flac_last_error = (C.c_char_p, [])
    w2l_selftest_wave_ops = (_i, [_p, _p, _p])
    w2l_batch_target_size = (_i, [_i, _i, _i, _p, _p, _p])
    w2l_batch_ctc_target_size = (_i, [_i, _i, _i, _p, _p, _p])
    w2l_fcc_workspace_size = (_sz, [_i, _i, _i])
    w2l_fcc_forward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p])
    w2l_fcc_backward = (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p])
    w2l_fac_workspace_size = (_sz, [_i, _i, _i, _i])
    w2l_fac_forward = (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p])
    w2l_fac_backward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p])
    w2l_asg_workspace_size = (_sz, [_i, _i, _i, _i])
    w2l_asg_forward = (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p])
    w2l_asg_backward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p])
    w2l_fac_viterbi = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p])
    w2l_fcc_range_flags = (_i, [_i, _i, _i, _p, _p, _p])
    w2l_fac_range_flags = (_i, [_i, _i, _i, _i, _p, _p, _p])
    w2l_linear_target = (_i, [_i, _i, _i, _p, _p, _p])
    w2l_fac_fullpath_forward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p])
    w2l_fac_fullpath_backward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p])
    w2l_viterbi_workspace_size = (_sz, [_i, _i, _i])
    w2l_viterbi_compute = (_i, [_i, _i, _i, _p, _p, _p, _p, _p])
    w2l_ctc_workspace_size = (_sz, [_i, _i, _i, _i])
    w2l_ctc_forward = (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p])
    w2l_ctc_backward = (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p])
    w2l_ctc_viterbi = (_i, [_i, _i, _i, _p, _p, _p])
    w2l_gemm_f32 = (_i, [_i, _i, _i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _i, _i, _p])
    w2l_linear_forward = (_i, [_i, _i, _i, _p, _p, _p, _p, _i, _p])
    w2l_linear_backward_data = (_i, [_i, _i, _i, _p, _p, _p, _i, _p, _f, _p])
    w2l_linear_backward_weight = (_i, [_i, _i, _i, _p, _p, _p, _p])
    w2l_linear_backward_weight_bias = (_i, [_i, _i, _i, _p, _p, _p, _p, _p])
    w2l_colsum = (_i, [_p, _p, _sz, _i, _p])
    w2l_set_matmul_precision = (_i, [_i])
    w2l_bf16_convert = (_i, [_p, _sz, _i, _sz, _p, _sz, _p, _sz, _p])
    w2l_bf16_convert_multi = (_i, [_i, _p, _p])
    w2l_gemm_bf16_images = (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _f, _p])
    w2l_bf16_convert_dropout = (_i, [_p, _sz, _i, _sz, _p, _sz, _p, _sz, _d, _u32, _u32, _p])
    w2l_gemm_bf16 = (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p])
    w2l_gemm_bf16_ex = (_i, [_i, _i, _i, _p, _i, _i, _p, _i, _i, _p, _i, _p, _i, _p, _p])
    w2l_gemm_bf16_grouped = (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p])
    w2l_tds_conv_bf16_image_elems = (_sz, [_p])
    w2l_tds_conv_bf16_prepare = (_i, [_p, _p, _p, _p, _p])
    w2l_tds_conv_bf16_forward = (_i, [_p, _p, _p, _p, _p, _i, _p])
    w2l_tds_conv_bf16_backward_data = (_i, [_p, _p, _p, _p, _p, _p])
    w2l_tds_conv_bf16_backward_filter = (_i, [_p, _p, _p, _p, _p])
    w2l_tds_conv_bf16_backward_filter_bias = (_i, [_p, _p, _p, _p, _p, _p])
    w2l_mfsc_spectrum = (_i, [_p, _p, _sz, _i, _i, _i, _p])
    w2l_mfsc_log_transpose = (_i, [_p, _p, _i, _i, _i, _i, _f, _p])
    w2l_weightnorm_forward = (_i, [_p, _p, _p, _p, _i, _i, _p])
    w2l_weightnorm_backward = (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p])
    w2l_conv_out_len = (_i, [_i, _i, _i, _i, _i])
    w2l_conv_same_pad = (_i, [_i, _i, _i])
    w2l_conv_forward = (_i, [_p, _p, _p, _p, _p, _i, _p])
    w2l_conv_backward_data = (_i, [_p, _p, _p, _p, _i, _p])
    w2l_conv_backward_data_add = (_i, [_p, _p, _p, _p, _p, _p])
    w2l_conv_backward_filter = (_i, [_p, _p, _p, _p, _p, _p])
    w2l_layernorm_scratch_doubles = (_sz, [_i, _sz])
    w2l_residual_layernorm_forward_images = (_i, [_i, _sz, _p, _p, _p, _p, _p, _f, _d, _u32, _u32, _p, _p, _p])
    w2l_layernorm_backward_images = (_i, [_i, _sz, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _d, _u32, _u32, _p])
    w2l_residual_layernorm_forward = (_i, [_i, _sz, _p, _p, _p, _p, _p, _f, _d, _u32, _u32, _p, _p, _p])
    w2l_layernorm_backward = (_i, [_i, _sz, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p])
    w2l_dropout_inplace = (_i, [_p, _sz, _d, _u32, _u32, _p])
    w2l_dropout_copy = (_i, [_p, _p, _sz, _d, _u32, _u32, _p])
    w2l_linear_forward_dropout = (_i, [_i, _i, _i, _p, _p, _p, _p, _i, _d, _u32, _u32, _p])
    w2l_linear_forward_dropout_add = (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _d, _u32, _u32, _p])
    w2l_linear_backward_data_add = (_i, [_i, _i, _i, _p, _p, _p, _p, _p])
    w2l_mask_backward = (_i, [_p, _p, _p, _sz, _f, _p])
    w2l_hexpand_forward = (_i, [_p, _p, _sz, _i, _i, _i, _i, _p])
    w2l_hexpand_backward = (_i, [_p, _p, _sz, _i, _i, _i, _i, _p])
    w2l_bgemm_f32 = (_i, [_p, _p, _p, _p, _p])
    w2l_bgemm_bf16 = (_i, [_p, _p, _p, _p, _p])
    w2l_attn_softmax_forward = (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p])
    w2l_attn_fused_forward = (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p])
    w2l_attn_fused_backward_workspace = (_sz, [_p, _i])
    w2l_attn_fused_forward_images = (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p])
    w2l_attn_fused_backward_images = (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p])
    w2l_attn_fused_backward = (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p])
    w2l_attn_key_lengths = (_i, [_p, _i, _i, _i, _p, _p])
    w2l_attn_key_lengths_full = (_i, [_p, _p, _i, _i, _i, _p, _p])
    w2l_attn_softmax_backward = (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p])
    w2l_pool_time_forward = (_i, [_p, _p, _i, _i, _i, _i, _i, _p])
    w2l_pool_time_backward = (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p])
    w2l_axpy = (_i, [_p, _p, _sz, _f, _p])
    w2l_transpose = (_i, [_p, _p, _i, _i, _i, _p])
    w2l_glu_forward = (_i, [_p, _p, _sz, _i, _p])
    w2l_glu_backward = (_i, [_p, _p, _p, _sz, _i, _p])
    w2l_sumsq = (_i, [_p, _sz, _p, _i, _p])
    w2l_specaugment_inplace = (_i, [_p, _i, _i, _i, _i, _i, _i, _f, _i, _u32, _p])
    w2l_fill = (_i, [_p, _sz, _f, _p])
    w2l_profile_enable = (_i, [_i])
    w2l_profile_report_kind = (_i, [_i, _p, _p, _p])
    w2l_profile_launches = (_i, [_i, _i, _p, _p, _p])
    w2l_sgd_step = (_i, [_p, _p, _p, _sz, _f, _f, _f, _f, _p, _p])
    w2l_sgd_step_guarded = (_i, [_p, _p, _p, _sz, _f, _f, _f, _f, _p, _p])
    w2l_adagrad_step_guarded = (_i, [_p, _p, _p, _sz, _f, _f, _f, _f, _p, _p])
    w2l_adadelta_step_guarded = (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _p, _p])
    w2l_grad_guard = (_i, [_p, _p, _i, _p])


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "H", "Cin", "Cout", "kw", "stride", "padl", "padr")]


class GemmEpilogue(C.Structure):  # w2l_gemm_epilogue
    _fields_ = [("mask", C.c_void_p), ("maskScale", C.c_float), ("addend", C.c_void_p), ("accumulate", C.c_int),
                ("dropP", C.c_double), ("dropSeed", C.c_uint32), ("dropStream", C.c_uint32)]


class BgemmDesc(C.Structure):  # w2l_bgemm_desc
    _fields_ = ([(n, C.c_int) for n in ("M", "N", "K", "G1", "G2")] +
                [(n, C.c_longlong) for n in ("sam", "sak", "a1", "a2", "sbk", "sbn", "b1", "b2", "ldc", "c1", "c2")] +
                [("accumulate", C.c_int), ("bandMode", C.c_int), ("bandT", C.c_int), ("bandH", C.c_int), ("bandOff", C.c_int)])


class Bf16ConvertDesc(C.Structure):  # w2l_bf16_convert_desc
    _fields_ = [("x", C.c_void_p), ("rows", C.c_size_t), ("cols", C.c_int), ("ldx", C.c_size_t), ("rowMajor", C.c_void_p),
                ("ldRows", C.c_size_t), ("transposed", C.c_void_p), ("ldTrans", C.c_size_t)]


class Bf16ImageSink(C.Structure):  # w2l_bf16_image_sink
    _fields_ = [("rowMajor", C.c_void_p), ("ldRows", C.c_size_t), ("transposed", C.c_void_p), ("ldTrans", C.c_size_t)]


class AttnFusedDesc(C.Structure):  # w2l_attn_fused_desc
    _fields_ = ([(n, C.c_int) for n in ("B", "H", "T", "d", "ld", "ldc", "W", "n0", "rlo")] + [("scale", C.c_float), ("dropP", C.c_double),
                ("dropSeed", C.c_uint32), ("dropStream", C.c_uint32)])


def check(status, what=""):
    if status == W2L_OK:
        return
    msg = f"{what}: {_ERR.get(status, status)}"
    if status == W2L_EHIP:
        msg += f" (hipError {lib().w2l_last_hip_error()})"
    if status == W2L_EINVAL:
        raise W2LInvalidArgument(msg)
    raise W2LError(msg)


def exported_symbols():
    """names declared in include/w2l_hip.h (parsed), for the ABI-completeness test"""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "w2l_hip.h")
    src = open(hdr).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(w2l_[a-z0-9_]+)\s*\(", src)))
