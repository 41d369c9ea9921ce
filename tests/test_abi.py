"""CPU-side checks: the C-ABI library loads without a GPU and exports every
symbol include/w2l_hip.h declares; host-side argument validation."""
import ctypes
import os

import pytest


def test_library_loads_and_exports_all_declared_symbols():
    from wav2letter_amd import _lib
    lib = _lib.lib()
    assert b"gfx950" in lib.w2l_version()
    names = _lib.exported_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/w2l_hip.h but not exported"


def test_workspace_queries_are_pure_host_functions():
    from wav2letter_amd import _lib
    lib = _lib.lib()
    assert lib.w2l_fcc_workspace_size(64, 2000, 30) >= 2 * 64 * 2000 * 30 * 4
    assert lib.w2l_fac_workspace_size(64, 2000, 30, 300) >= 64 * 2000 * 300 * 4
    assert lib.w2l_viterbi_workspace_size(64, 2000, 30) >= 64 * 2000 * 30
    assert lib.w2l_ctc_workspace_size(32, 188, 9998, 80) > 0
    assert lib.w2l_fcc_workspace_size(0, 10, 10) == 0
    # ASG in one call: room for both criteria's workspaces, the target sizes, the second loss and the gradient scratch
    asg = lib.w2l_asg_workspace_size(64, 2000, 30, 300)
    assert asg >= lib.w2l_fcc_workspace_size(64, 2000, 30) + lib.w2l_fac_workspace_size(64, 2000, 30, 300) + 64 * 2000 * 30 * 4
    assert lib.w2l_asg_workspace_size(64, 2000, 30, 0) == 0 and lib.w2l_asg_workspace_size(0, 2000, 30, 300) == 0


def test_null_and_bad_shapes_are_rejected_without_touching_the_gpu():
    from wav2letter_amd import _lib
    lib = _lib.lib()
    assert lib.w2l_fcc_forward(0, 10, 10, 0, None, None, None, None, None, None) == _lib.W2L_EINVAL
    assert lib.w2l_fcc_forward(2, 10, 10, 0, None, None, None, None, None, None) == _lib.W2L_EINVAL
    assert lib.w2l_ctc_forward(2, 10, 1, 4, 0, None, None, None, None, None, None) == _lib.W2L_EINVAL
    assert lib.w2l_viterbi_compute(1, 1, 0, None, None, None, None, None) == _lib.W2L_EINVAL
    assert lib.w2l_asg_forward(2, 10, 5, 3, 0, None, None, None, None, None, None) == _lib.W2L_EINVAL
    assert lib.w2l_asg_backward(2, 10, 5, 0, None, None, None, None, None, None, None) == _lib.W2L_EINVAL


def test_scale_mode_mapping():
    from wav2letter_amd import CriterionScaleMode, getCriterionScaleMode
    from wav2letter_amd._lib import W2LInvalidArgument
    assert getCriterionScaleMode("target", True) == CriterionScaleMode.TARGET_SZ_SQRT
    assert getCriterionScaleMode("target", False) == CriterionScaleMode.TARGET_SZ
    assert getCriterionScaleMode("input", True) == CriterionScaleMode.INPUT_SZ_SQRT
    assert getCriterionScaleMode("none", True) == CriterionScaleMode.NONE
    with pytest.raises(W2LInvalidArgument):
        getCriterionScaleMode("bogus", False)


def test_cpu_tensors_fail_loudly():
    import torch
    from wav2letter_amd import CTCLoss
    from wav2letter_amd._lib import W2LError
    with pytest.raises(W2LError):
        CTCLoss()(torch.zeros(1, 4, 5), torch.zeros(1, 2, dtype=torch.int32))
