"""Python front-end of the C++ trainer (wav2letter_amd/csrc/host/trainer.cpp).

Mirrors the reference's Trainer step (recipes/slimIPL/src/Train.cpp:1454-1804): torch is
used to own the device arenas and to run the ONE gradient all-reduce per step over RCCL
(torch.distributed backend "nccl"); every kernel launch happens inside libw2l_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .criterion import CriterionScaleMode

_sigs_done = set()   # ids of the CDLL objects whose trainer signatures are declared (product / probe library)


def _lib_tr():
    L = _lib.lib()
    if id(L) not in _sigs_done:
        vp, i, sz, f, d, u32, u64 = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double, C.c_uint32, C.c_uint64
        L.w2l_host_last_error.restype = C.c_char_p
        L.w2l_trainer_create.restype = vp
        L.w2l_trainer_create.argtypes = [C.c_char_p, i, i, C.c_char_p, i, d]
        L.w2l_trainer_destroy.argtypes = [vp]
        L.w2l_trainer_describe.restype = C.c_char_p
        L.w2l_trainer_describe.argtypes = [vp]
        for n in ("w2l_trainer_param_floats", "w2l_trainer_net_param_floats", "w2l_trainer_grad_floats"):
            getattr(L, n).restype = sz
            getattr(L, n).argtypes = [vp]
        L.w2l_trainer_num_params.argtypes = [vp]
        L.w2l_trainer_param_info.argtypes = [vp, i, C.c_char_p, i, C.POINTER(sz), C.POINTER(sz)]
        L.w2l_trainer_init_params.argtypes = [vp, vp, u64]
        L.w2l_trainer_import_param.argtypes = [vp, i, vp, vp]
        L.w2l_trainer_export_param.argtypes = [vp, i, vp, vp]
        L.w2l_trainer_plan.argtypes = [vp, i, i, i, C.POINTER(sz), C.POINTER(sz), C.POINTER(i)]
        L.w2l_trainer_bind.argtypes = [vp, vp, vp, vp, vp, vp]
        L.w2l_trainer_forward.argtypes = [vp, vp, i, C.POINTER(vp), vp]
        L.w2l_trainer_forward_backward.argtypes = [vp, vp, vp, C.POINTER(vp), vp]
        L.w2l_trainer_backward.argtypes = [vp, vp, vp]
        L.w2l_trainer_update.argtypes = [vp, f, f, f, f, f, i, vp]
        L.w2l_trainer_viterbi.argtypes = [vp, vp, vp, vp]
        L.w2l_trainer_set_step.argtypes = [vp, u32]
        L.w2l_trainer_set_mixed_precision.argtypes = [vp, i]
        L.w2l_trainer_set_optimizer.argtypes = [vp, i, i]
        L.w2l_trainer_bind_state2.argtypes = [vp, vp]
        L.w2l_trainer_set_input_sizes.argtypes = [vp, vp]
        L.w2l_trainer_set_linseg.argtypes = [vp, u32]
        L.w2l_trainer_grad_norm.argtypes = [vp, C.POINTER(C.c_double), vp]
        L.w2l_trainer_skipped_updates.argtypes = [vp, C.POINTER(u64), vp]
        L.w2l_trainer_set_grad_buckets.argtypes = [vp, i, C.POINTER(sz)]
        L.w2l_trainer_wait_bucket.argtypes = [vp, i, vp]
        L.w2l_arch_check.argtypes = [C.c_char_p, i, i, C.POINTER(i)]
        L.w2l_flags_check.argtypes = [C.c_char_p, C.POINTER(i)]
        _sigs_done.add(id(L))
    return L


def _check(st, what):
    if st != 0:
        msg = _lib_tr().w2l_host_last_error().decode()
        if st == _lib.W2L_EINVAL:
            raise _lib.W2LInvalidArgument(f"{what}: {msg}")
        raise _lib.W2LError(f"{what}: {msg}")


def arch_check(arch_text, nfeat, nlabel):
    """parse an arch file (all tokens of the reference grammar); returns the number of layer lines"""
    n = C.c_int(0)
    _check(_lib_tr().w2l_arch_check(arch_text.encode(), nfeat, nlabel, C.byref(n)), "arch")
    return n.value


def flags_check(flags_text):
    n = C.c_int(0)
    _check(_lib_tr().w2l_flags_check(flags_text.encode(), C.byref(n)), "flags")
    return n.value


class Trainer:
    def __init__(self, arch_text, nfeat, nlabel, criterion="ctc", scalemode=CriterionScaleMode.NONE,
                 transdiag=0.0, device="cuda"):
        L = _lib_tr()
        self.L = L
        self.h = L.w2l_trainer_create(arch_text.encode(), nfeat, nlabel, criterion.encode(), int(scalemode),
                                      float(transdiag))
        if not self.h:
            raise _lib.W2LInvalidArgument(L.w2l_host_last_error().decode())
        self.nfeat, self.nlabel = nfeat, nlabel
        self.criterion = criterion
        self._pending_mom = None
        self.device = device
        self.n_floats = L.w2l_trainer_param_floats(self.h)
        self.n_net = L.w2l_trainer_net_param_floats(self.h)
        self.host_params = np.zeros(self.n_floats, np.float32)
        self.params = self.grads = self.mom = None
        self.B = self.T = self.Lt = self.Tout = 0

    def __del__(self):
        try:
            if self.h:
                self.L.w2l_trainer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def describe(self):
        return self.L.w2l_trainer_describe(self.h).decode()

    def param_table(self):
        out = []
        for i in range(self.L.w2l_trainer_num_params(self.h)):
            name = C.create_string_buffer(64)
            n, off = C.c_size_t(0), C.c_size_t(0)
            self.L.w2l_trainer_param_info(self.h, i, name, 64, C.byref(n), C.byref(off))
            out.append((name.value.decode(), n.value, off.value))
        return out

    def init_params(self, seed=0):
        _check(self.L.w2l_trainer_init_params(self.h, self.host_params.ctypes.data, seed), "init_params")

    def import_param(self, i, ref):
        ref = np.ascontiguousarray(ref, np.float32)
        _check(self.L.w2l_trainer_import_param(self.h, i, ref.ctypes.data, self.host_params.ctypes.data), "import")

    def export_from(self, i, host_arena):
        name, n, off = self.param_table()[i]
        out = np.zeros(n, np.float32)
        host_arena = np.ascontiguousarray(host_arena, np.float32)
        _check(self.L.w2l_trainer_export_param(self.h, i, host_arena.ctypes.data, out.ctypes.data), "export")
        return out

    def to_device(self):
        """upload host_params, allocate grads / momentum"""
        self.params = torch.from_numpy(self.host_params).to(self.device)
        # gradient arena = parameters + a 4-float tail: tail[0] carries this rank's batch size through the SAME
        # all-reduce as the gradients (Train.cpp:1743-1747 reduces it separately)
        self.grads_full = torch.zeros(self.L.w2l_trainer_grad_floats(self.h), dtype=torch.float32, device=self.device)
        self.grads = self.grads_full[:self.n_floats]
        self.mom = torch.zeros_like(self.params)
        if getattr(self, "_pending_mom", None) is not None:   # checkpoint.load() before to_device()
            self.mom.copy_(torch.from_numpy(self._pending_mom))
            self._pending_mom = None
        self._bind_state2()
        if self.B:
            self._bind()

    def _bind_state2(self):
        if self.params is not None and "adadelta" in getattr(self, "_optim", ()) and getattr(self, "state2", None) is None:
            self.state2 = torch.zeros_like(self.params)            # Adadelta's accDelta (accGrad lives in self.mom)
            if getattr(self, "_pending_state2", None) is not None:  # checkpoint.load() before to_device()
                self.state2.copy_(torch.from_numpy(self._pending_state2))
                self._pending_state2 = None
            _check(self.L.w2l_trainer_bind_state2(self.h, self.state2.data_ptr()), "bind_state2")

    def plan(self, B, T, L):
        af, cw, to = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        _check(self.L.w2l_trainer_plan(self.h, B, T, L, C.byref(af), C.byref(cw), C.byref(to)), "plan")
        self.B, self.T, self.Lt, self.Tout = B, T, L, to.value
        self.arena = torch.empty(af.value, dtype=torch.float32, device=self.device)
        self.crit_ws = torch.empty(max(cw.value, 256), dtype=torch.uint8, device=self.device)
        if self.params is not None:
            self._bind()
        return to.value

    def _bind(self):
        _check(self.L.w2l_trainer_bind(self.h, self.params.data_ptr(), self.grads_full.data_ptr(), self.mom.data_ptr(),
                                       self.arena.data_ptr(), self.crit_ws.data_ptr()), "bind")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _check_input(self, x, target=None):
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                and tuple(x.shape) == (self.B, self.nfeat, self.T)):
            raise _lib.W2LInvalidArgument(
                f"input must be a contiguous float32 CUDA tensor [B={self.B}][NFEAT={self.nfeat}][T={self.T}], got "
                f"{tuple(x.shape) if torch.is_tensor(x) else type(x)}")
        if target is not None and not (torch.is_tensor(target) and target.is_cuda and target.dtype == torch.int32
                                       and target.is_contiguous() and tuple(target.shape) == (self.B, self.Lt)):
            raise _lib.W2LInvalidArgument(f"target must be a contiguous int32 CUDA tensor [B={self.B}][L={self.Lt}]")

    def forward(self, x, train=False):
        """x: [B][NFEAT][T] float32 -> emissions view [B][T'][N] (aliases the arena)"""
        self._check_input(x)
        ptr = C.c_void_p(0)
        _check(self.L.w2l_trainer_forward(self.h, x.data_ptr(), int(train), C.byref(ptr), self._stream()), "forward")
        off = (ptr.value - self.arena.data_ptr()) // 4
        return self.arena[off:off + self.B * self.Tout * self.nlabel].view(self.B, self.Tout, self.nlabel)

    def forward_backward(self, x, target):
        self._check_input(x, target)
        ptr = C.c_void_p(0)
        _check(self.L.w2l_trainer_forward_backward(self.h, x.data_ptr(), target.data_ptr(), C.byref(ptr),
                                                   self._stream()), "forward_backward")
        off = (ptr.value - self.arena.data_ptr()) // 4
        return self.arena[off:off + self.B]

    def backward(self, d_emission):
        """the network's backward pass alone from a caller-supplied gradient of the emissions [B][T'][N] (after
        forward(train=True)): the fl::Module boundary for a binder that keeps its own criterion"""
        assert d_emission.is_cuda and d_emission.dtype == torch.float32 and d_emission.is_contiguous()
        assert d_emission.numel() == self.B * self.Tout * self.nlabel, (tuple(d_emission.shape), self.B, self.Tout, self.nlabel)
        _check(self.L.w2l_trainer_backward(self.h, d_emission.data_ptr(), self._stream()), "backward")

    def update(self, lr, lrcrit=0.0, momentum=0.0, max_grad_norm=0.0, total_batch=None, clamp_crit=True):
        """total_batch: None -> this rank's B; a number -> that; 0 (or "reduced") -> the all-reduced batch size that
        rode in the gradient arena's tail (data-parallel runs)"""
        if total_batch == "reduced":
            total_batch = 0
        tb = float(total_batch if total_batch is not None else self.B)
        _check(self.L.w2l_trainer_update(self.h, lr, lrcrit, momentum, max_grad_norm, tb, int(clamp_crit),
                                         self._stream()), "update")

    def viterbi(self, emission):
        path = torch.empty(self.B, self.Tout, dtype=torch.int32, device=self.device)
        _check(self.L.w2l_trainer_viterbi(self.h, emission.data_ptr(), path.data_ptr(), self._stream()), "viterbi")
        return path

    def grad_norm(self):
        """gradient norm of the last update (synchronises); non-finite => that update was skipped on every rank"""
        n = C.c_double(0.0)
        _check(self.L.w2l_trainer_grad_norm(self.h, C.byref(n), self._stream()), "grad_norm")
        return n.value

    def skipped_updates(self):
        """updates skipped because the reduced gradient was non-finite (synchronises)"""
        n = C.c_uint64(0)
        _check(self.L.w2l_trainer_skipped_updates(self.h, C.byref(n), self._stream()), "skipped_updates")
        return n.value

    def set_mixed_precision(self, on=True):
        """bf16 multiplies (fp32 accumulate / storage / master weights) in the network's fl::Linear GEMMs"""
        _check(self.L.w2l_trainer_set_mixed_precision(self.h, int(bool(on))), "mixed precision")

    def set_optimizer(self, netoptim="sgd", critoptim="sgd"):
        """--netoptim / --critoptim of the reference Trainer (Train.cpp:577-582): "sgd" (momentum) "adagrad" or "adadelta" (rho 0.9, eps 1e-8:
        the recipe of BASELINE config 5).  Adagrad keeps its squared-gradient sums in the momentum arena; Adadelta its accGrad
        there and accDelta in a second arena (self.state2)"""
        kinds = {"sgd": 0, "adagrad": 1, "adadelta": 2}
        _check(self.L.w2l_trainer_set_optimizer(self.h, kinds[netoptim], kinds[critoptim]), "set_optimizer")
        self._optim = (netoptim, critoptim)
        self._bind_state2()

    def set_input_sizes(self, sizes):
        """per-utterance input sizes of the next batches (float32 CUDA tensor [B] in any unit, or None): the Transformer blocks
        mask the padded keys as forwardSequentialModuleWithPadMask does (cpc/SequentialBuilder.cpp:58-81); the tensor is kept
        alive by the trainer and read at every forward"""
        if sizes is not None:
            if not (torch.is_tensor(sizes) and sizes.dtype == torch.float32 and sizes.is_cuda and sizes.is_contiguous()
                    and sizes.numel() == self.B):
                raise ValueError("input sizes: contiguous float32 CUDA tensor of B elements")
        self._input_sizes = sizes
        _check(self.L.w2l_trainer_set_input_sizes(self.h, sizes.data_ptr() if sizes is not None else None), "set_input_sizes")

    def set_step(self, step):
        self.L.w2l_trainer_set_step(self.h, step)

    def set_linseg(self, updates):
        """--linseg=n: LinSegCriterion for the first n updates of an ASG run (call before plan())"""
        _check(self.L.w2l_trainer_set_linseg(self.h, int(updates)), "linseg")

    # ---- data-parallel overlap (parallel.OverlappedReducer drives these)
    def set_grad_buckets(self, offsets):
        """bucket k = grads[offsets[k]:offsets[k+1]] (last one to the end); forward_backward then records
        one event per bucket as soon as that part of the gradient arena is final"""
        arr = (C.c_size_t * len(offsets))(*offsets)
        _check(self.L.w2l_trainer_set_grad_buckets(self.h, len(offsets), arr), "set_grad_buckets")
        self.bucket_offsets = list(offsets)

    def wait_bucket(self, k, stream):
        """make `stream` (torch.cuda.Stream) wait until bucket k of the step enqueued last is final"""
        _check(self.L.w2l_trainer_wait_bucket(self.h, k, stream.cuda_stream), "wait_bucket")
