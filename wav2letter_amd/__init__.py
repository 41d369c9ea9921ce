"""wav2letter_amd -- MI355X (gfx950) native wav2letter acoustic-training hot path.

All compute runs in libw2l_hip.so (hand-written HIP kernels behind the C ABI in
include/w2l_hip.h).  torch is used for device memory, streams and
torch.distributed (RCCL) only.
"""
from . import _lib  # noqa: F401
from . import text  # noqa: F401  (token dictionary, lexicon, targets, TER / WER remap: host logic)
from .criterion import (ASGLoss, CTCLoss, CriterionScaleMode, ForceAlignmentCriterion,  # noqa: F401
                        FullConnectionCriterion, LinSegCriterion, SequenceCriterion, getCriterionScaleMode, linear_target)
