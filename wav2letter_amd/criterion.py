"""Host-side mirror of fl::pkg::speech sequence criteria over the C ABI.

Mirrors (same names, argument meaning, error behaviour) the interface the
reference's Trainer uses (recipes/slimIPL/src/Train.cpp:406-410, :1675, :838;
shape of SequenceCriterion in recipes/joint_training_vox_populi/cpc/CPCCriterion.h:30-50):

  ASGLoss(N, scalemode, transdiag).forward(emission, target) -> loss[B]
  CTCLoss(scalemode).forward(emission, target) -> loss[B]
  crit.viterbiPath(emission) -> int32 [B][T];  crit.viterbiPathWithTarget(...)

Tensors are torch CUDA(HIP) tensors used only as device buffers: emission is
[B][T][N] float32 contiguous (== ArrayFire dims (N,T,B)), target [B][L] int32
padded with -1.  Every op runs in libw2l_hip.so; there is no PyTorch fallback.
"""
import enum

import torch

from . import _lib


class CriterionScaleMode(enum.IntEnum):
    NONE = 0
    INPUT_SZ = 1
    INPUT_SZ_SQRT = 2
    TARGET_SZ = 3
    TARGET_SZ_SQRT = 4


def getCriterionScaleMode(onorm: str, sqnorm: bool) -> CriterionScaleMode:
    """recipes/slimIPL/src/Train.cpp:389 (--onorm / --sqnorm)"""
    if onorm == "none":
        return CriterionScaleMode.NONE
    if onorm == "input":
        return CriterionScaleMode.INPUT_SZ_SQRT if sqnorm else CriterionScaleMode.INPUT_SZ
    if onorm == "target":
        return CriterionScaleMode.TARGET_SZ_SQRT if sqnorm else CriterionScaleMode.TARGET_SZ
    raise _lib.W2LInvalidArgument(f"invalid onorm option: {onorm}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.W2LError("w2l criteria run on the GPU only (no CPU fallback)")


def _emission_checks(emission, target=None):
    if emission.dim() != 3:
        raise _lib.W2LInvalidArgument("emission must be [B][T][N]")
    if emission.dtype != torch.float32:
        raise _lib.W2LInvalidArgument("emission must be float32")
    if target is not None:
        if target.dtype != torch.int32:
            raise _lib.W2LInvalidArgument("target must be int32")
        if target.dim() != 2 or target.shape[0] != emission.shape[0]:
            raise _lib.W2LInvalidArgument("target must be [B][L]")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def batch_target_size(target, max_size, ctc=False):
    B, L = target.shape
    out = torch.empty(B, dtype=torch.int32, device=target.device)
    fn = _lib.lib().w2l_batch_ctc_target_size if ctc else _lib.lib().w2l_batch_target_size
    _lib.check(fn(B, L, int(max_size), target.data_ptr(), out.data_ptr(), _stream()), "batch_target_size")
    return out


class _FCC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emission, trans, target_size, scale_mode):
        L = _lib.lib()
        emission = emission.contiguous()
        trans = trans.contiguous()
        B, T, N = emission.shape
        ws = _ws(L.w2l_fcc_workspace_size(B, T, N), emission.device)
        loss = torch.empty(B, dtype=torch.float32, device=emission.device)
        _lib.check(L.w2l_fcc_forward(B, T, N, int(scale_mode), emission.data_ptr(), target_size.data_ptr(),
                                     trans.data_ptr(), loss.data_ptr(), ws.data_ptr(), _stream()), "fcc_forward")
        ctx.save_for_backward(trans, ws)
        ctx.dims = (B, T, N)
        _FCC.last = (ws, (B, T, N))   # for range_flags(): which utterances took the log-domain path (diagnostics)
        return loss

    @staticmethod
    def backward(ctx, grad):
        L = _lib.lib()
        trans, ws = ctx.saved_tensors
        B, T, N = ctx.dims
        grad = grad.contiguous().float()
        dx = torch.empty(B, T, N, dtype=torch.float32, device=grad.device)
        dt = torch.empty(N, N, dtype=torch.float32, device=grad.device)
        _lib.check(L.w2l_fcc_backward(B, T, N, trans.data_ptr(), grad.data_ptr(), dx.data_ptr(), dt.data_ptr(),
                                      ws.data_ptr(), _stream()), "fcc_backward")
        return dx, dt, None, None


class _FAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emission, trans, target, target_size, scale_mode):
        L = _lib.lib()
        emission = emission.contiguous()
        trans = trans.contiguous()
        target = target.contiguous()
        B, T, N = emission.shape
        Lt = target.shape[1]
        ws = _ws(L.w2l_fac_workspace_size(B, T, N, Lt), emission.device)
        loss = torch.empty(B, dtype=torch.float32, device=emission.device)
        _lib.check(L.w2l_fac_forward(B, T, N, Lt, int(scale_mode), emission.data_ptr(), target.data_ptr(),
                                     target_size.data_ptr(), trans.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                     _stream()), "fac_forward")
        ctx.save_for_backward(target, target_size, ws)
        ctx.dims = (B, T, N, Lt)
        _FAC.last = (ws, (B, T, N, Lt))
        return loss

    @staticmethod
    def backward(ctx, grad):
        L = _lib.lib()
        target, target_size, ws = ctx.saved_tensors
        B, T, N, Lt = ctx.dims
        grad = grad.contiguous().float()
        dx = torch.empty(B, T, N, dtype=torch.float32, device=grad.device)
        dt = torch.empty(N, N, dtype=torch.float32, device=grad.device)
        _lib.check(L.w2l_fac_backward(B, T, N, Lt, target.data_ptr(), target_size.data_ptr(), grad.data_ptr(),
                                      dx.data_ptr(), dt.data_ptr(), ws.data_ptr(), _stream()), "fac_backward")
        return dx, dt, None, None, None


class _ASG(torch.autograd.Function):
    """ASGLoss = FullConnectionCriterion - ForceAlignmentCriterion in one call each way (w2l_asg_forward / w2l_asg_backward: the two
    criteria side by side on the current stream and a library-owned side stream, the launch sequence of the C++ criterion)"""

    @staticmethod
    def forward(ctx, emission, trans, target, scale_mode):
        L = _lib.lib()
        emission = emission.contiguous()
        trans = trans.contiguous()
        target = target.contiguous()
        B, T, N = emission.shape
        Lt = target.shape[1]
        nbytes = L.w2l_asg_workspace_size(B, T, N, Lt)
        if not nbytes:
            raise _lib.W2LError("w2l_asg_workspace_size: unsupported shape")
        ws = _ws(nbytes, emission.device)
        loss = torch.empty(B, dtype=torch.float32, device=emission.device)
        _lib.check(L.w2l_asg_forward(B, T, N, Lt, int(scale_mode), emission.data_ptr(), target.data_ptr(), trans.data_ptr(),
                                     loss.data_ptr(), ws.data_ptr(), _stream()), "asg_forward")
        ctx.save_for_backward(target, trans, ws)
        ctx.dims = (B, T, N, Lt)
        return loss

    @staticmethod
    def backward(ctx, grad):
        L = _lib.lib()
        target, trans, ws = ctx.saved_tensors
        B, T, N, Lt = ctx.dims
        grad = grad.contiguous().float()
        dx = torch.empty(B, T, N, dtype=torch.float32, device=grad.device)
        dt = torch.empty(N, N, dtype=torch.float32, device=grad.device)
        _lib.check(L.w2l_asg_backward(B, T, N, Lt, target.data_ptr(), trans.data_ptr(), grad.data_ptr(), dx.data_ptr(), dt.data_ptr(),
                                      ws.data_ptr(), _stream()), "asg_backward")
        return dx, dt, None, None


def fcc_range_flags():
    """int32 [B]: 1 where an utterance of the LAST FullConnectionCriterion forward left the range of the fp32 scaled-domain scan and
    was recomputed by the log-domain kernels (w2l_fcc_range_flags); results are exact either way"""
    ws, (B, T, N) = _FCC.last
    out = torch.empty(B, dtype=torch.int32, device=ws.device)
    _lib.check(_lib.lib().w2l_fcc_range_flags(B, T, N, ws.data_ptr(), out.data_ptr(), _stream()), "fcc_range_flags")
    return out


def fac_range_flags():
    """the same for the LAST ForceAlignmentCriterion forward (w2l_fac_range_flags)"""
    ws, (B, T, N, Lt) = _FAC.last
    out = torch.empty(B, dtype=torch.int32, device=ws.device)
    _lib.check(_lib.lib().w2l_fac_range_flags(B, T, N, Lt, ws.data_ptr(), out.data_ptr(), _stream()), "fac_range_flags")
    return out


class _FACFullPath(torch.autograd.Function):
    """ForceAlignmentCriterion on a length-T target: one alignment (w2l_fac_fullpath_*)"""

    @staticmethod
    def forward(ctx, emission, trans, path, scale_mode):
        L = _lib.lib()
        emission = emission.contiguous()
        trans = trans.contiguous()
        B, T, N = emission.shape
        loss = torch.empty(B, dtype=torch.float32, device=emission.device)
        _lib.check(L.w2l_fac_fullpath_forward(B, T, N, int(scale_mode), emission.data_ptr(), path.data_ptr(),
                                              trans.data_ptr(), loss.data_ptr(), _stream()), "fac_fullpath_forward")
        ctx.save_for_backward(path)
        ctx.dims = (B, T, N, int(scale_mode))
        return loss

    @staticmethod
    def backward(ctx, grad):
        L = _lib.lib()
        (path,) = ctx.saved_tensors
        B, T, N, mode = ctx.dims
        grad = grad.contiguous().float()
        dx = torch.empty(B, T, N, dtype=torch.float32, device=grad.device)
        dt = torch.empty(N, N, dtype=torch.float32, device=grad.device)
        _lib.check(L.w2l_fac_fullpath_backward(B, T, N, mode, path.data_ptr(), grad.data_ptr(), dx.data_ptr(),
                                               dt.data_ptr(), _stream()), "fac_fullpath_backward")
        return dx, dt, None, None


def linear_target(target, T):
    """Flashlight getLinearTarget: [B][L] labels (-1 padded) stretched to [B][T]"""
    _check_dev(target)
    target = target.contiguous()
    B, L = target.shape
    out = torch.empty(B, T, dtype=torch.int32, device=target.device)
    _lib.check(_lib.lib().w2l_linear_target(B, L, int(T), target.data_ptr(), out.data_ptr(), _stream()), "linear_target")
    return out


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emission, target, target_size, scale_mode):
        L = _lib.lib()
        emission = emission.contiguous()
        target = target.contiguous()
        B, T, N = emission.shape
        Lt = target.shape[1]
        ws = _ws(L.w2l_ctc_workspace_size(B, T, N, Lt), emission.device)
        loss = torch.empty(B, dtype=torch.float32, device=emission.device)
        _lib.check(L.w2l_ctc_forward(B, T, N, Lt, int(scale_mode), emission.data_ptr(), target.data_ptr(),
                                     target_size.data_ptr(), loss.data_ptr(), ws.data_ptr(), _stream()),
                   "ctc_forward")
        ctx.save_for_backward(emission, target, target_size, ws)
        ctx.dims = (B, T, N, Lt)
        return loss

    @staticmethod
    def backward(ctx, grad):
        L = _lib.lib()
        emission, target, target_size, ws = ctx.saved_tensors
        B, T, N, Lt = ctx.dims
        grad = grad.contiguous().float()
        dx = torch.empty(B, T, N, dtype=torch.float32, device=grad.device)
        _lib.check(L.w2l_ctc_backward(B, T, N, Lt, emission.data_ptr(), target.data_ptr(), target_size.data_ptr(),
                                      grad.data_ptr(), dx.data_ptr(), ws.data_ptr(), _stream()), "ctc_backward")
        return dx, None, None, None


class SequenceCriterion(torch.nn.Module):
    """fl::pkg::speech::SequenceCriterion: forward({emission,target}) -> {loss[B]},
    viterbiPath(emission) -> [B][T] int32."""

    def prettyString(self):
        return type(self).__name__


class FullConnectionCriterion(SequenceCriterion):
    def __init__(self, N, scalemode=CriterionScaleMode.NONE, transitions=None):
        super().__init__()
        self.N, self.scalemode = N, scalemode
        self.transitions = transitions if transitions is not None else torch.nn.Parameter(torch.zeros(N, N))

    def forward(self, emission, target):
        _emission_checks(emission, target)
        _check_dev(emission, target)
        if emission.shape[2] != self.N:
            raise _lib.W2LInvalidArgument("FullConnectionCriterion: N doesn't match with the letter size")
        ts = batch_target_size(target, emission.shape[1])
        return _FCC.apply(emission, self.transitions, ts, self.scalemode)


class ForceAlignmentCriterion(SequenceCriterion):
    def __init__(self, N, scalemode=CriterionScaleMode.NONE, transitions=None):
        super().__init__()
        self.N, self.scalemode = N, scalemode
        self.transitions = transitions if transitions is not None else torch.nn.Parameter(torch.zeros(N, N))

    def forward(self, emission, target):
        _emission_checks(emission, target)
        _check_dev(emission, target)
        if emission.shape[2] != self.N:
            raise _lib.W2LInvalidArgument("ForceAlignmentCriterion: N doesn't match with the letter size")
        ts = batch_target_size(target, emission.shape[1])
        return _FAC.apply(emission, self.transitions, target, ts, self.scalemode)

    def viterbiPath(self, emission, target):
        _emission_checks(emission, target)
        L = _lib.lib()
        emission = emission.contiguous()
        B, T, N = emission.shape
        Lt = target.shape[1]
        ts = batch_target_size(target, T)
        ws = _ws(L.w2l_fac_workspace_size(B, T, N, Lt), emission.device)
        path = torch.empty(B, T, dtype=torch.int32, device=emission.device)
        _lib.check(L.w2l_fac_viterbi(B, T, N, Lt, emission.data_ptr(), target.data_ptr(), ts.data_ptr(),
                                     self.transitions.detach().contiguous().data_ptr(), path.data_ptr(),
                                     ws.data_ptr(), _stream()), "fac_viterbi")
        return path


class ASGLoss(SequenceCriterion):
    """AutoSegmentationCriterion = FCC - FAC sharing one N x N transition
    parameter initialised to transdiag * I (recipes/slimIPL/src/Train.cpp:410;
    --transdiag, recipes/conv_glu/librispeech/train.cfg:25)."""

    def __init__(self, N, scalemode=CriterionScaleMode.NONE, transdiag=0.0):
        super().__init__()
        self.N, self.scalemode = N, scalemode
        self.transitions = torch.nn.Parameter(torch.eye(N) * float(transdiag))
        self.fac = ForceAlignmentCriterion(N, scalemode, self.transitions)
        self.fcc = FullConnectionCriterion(N, scalemode, self.transitions)
        self._side = None

    def forward(self, emission, target):
        # FCC and FAC are independent length-T serial scans that use B of the 256 CUs each: w2l_asg_forward / w2l_asg_backward run
        # them side by side (one call each way: the C++ criterion's launch sequence, fused for the letter-sized label sets)
        if not emission.is_cuda:
            return self.fcc(emission, target) - self.fac(emission, target)  # raises the reference's error
        _emission_checks(emission, target)
        _check_dev(emission, target)
        if emission.shape[2] != self.N:
            raise _lib.W2LInvalidArgument("ASGLoss: N doesn't match with the letter size")
        return _ASG.apply(emission, self.transitions, target, self.scalemode)

    def viterbiPath(self, emission, inputSize=None):
        _emission_checks(emission)
        _check_dev(emission)
        L = _lib.lib()
        emission = emission.contiguous()
        B, T, N = emission.shape
        ws = _ws(L.w2l_viterbi_workspace_size(B, T, N), emission.device)
        path = torch.empty(B, T, dtype=torch.int32, device=emission.device)
        _lib.check(L.w2l_viterbi_compute(B, T, N, emission.data_ptr(),
                                         self.transitions.detach().contiguous().data_ptr(), path.data_ptr(),
                                         ws.data_ptr(), _stream()), "viterbi_compute")
        return path

    def viterbiPathWithTarget(self, emission, target):
        return self.fac.viterbiPath(emission, target)

    def prettyString(self):
        return "AutoSegmentationCriterion"


class LinSegCriterion(SequenceCriterion):
    """LinearSegmentationCriterion: ASG on the target stretched linearly over the T frames, used for the first
    --linseg updates of every ASG recipe with the ASG criterion's own transition parameter
    (`linseg->setParams(criterion->param(0), 0)`, recipes/slimIPL/src/Train.cpp:592-596)."""

    def __init__(self, N, scalemode=CriterionScaleMode.NONE, transitions=None):
        super().__init__()
        self.N, self.scalemode = N, scalemode
        self.transitions = transitions if transitions is not None else torch.nn.Parameter(torch.zeros(N, N))

    def setParams(self, var, pos=0):
        if pos != 0:
            raise _lib.W2LInvalidArgument("LinSegCriterion has one parameter (transitions)")
        self.transitions = var

    def forward(self, emission, target):
        _emission_checks(emission, target)
        _check_dev(emission, target)
        if emission.shape[2] != self.N:
            raise _lib.W2LInvalidArgument("LinSegCriterion: N doesn't match with the letter size")
        T = emission.shape[1]
        lin = linear_target(target, T)
        ts = batch_target_size(lin, T)          # T, or 0 for a row that could not be stretched
        fcc = _FCC.apply(emission, self.transitions, ts, self.scalemode)
        fac = _FACFullPath.apply(emission, self.transitions, lin, self.scalemode)
        return fcc - fac

    def prettyString(self):
        return "LinearSegmentationCriterion"


class CTCLoss(SequenceCriterion):
    """ConnectionistTemporalClassificationCriterion(scalemode); blank = N-1"""

    def __init__(self, scalemode=CriterionScaleMode.NONE):
        super().__init__()
        self.scalemode = scalemode

    def forward(self, emission, target):
        _emission_checks(emission, target)
        _check_dev(emission, target)
        ts = batch_target_size(target, emission.shape[1], ctc=True)
        return _CTC.apply(emission, target, ts, self.scalemode)

    def viterbiPath(self, emission, inputSize=None):
        _emission_checks(emission)
        _check_dev(emission)
        emission = emission.contiguous()
        B, T, N = emission.shape
        path = torch.empty(B, T, dtype=torch.int32, device=emission.device)
        _lib.check(_lib.lib().w2l_ctc_viterbi(B, T, N, emission.data_ptr(), path.data_ptr(), _stream()),
                   "ctc_viterbi")
        return path

    def prettyString(self):
        return "ConnectionistTemporalClassificationCriterion"
