"""CPU: the index arithmetic of the fused attention backward (oracle/attention_bwd_model.py: tile layouts, reverse skew, global
window numbering, band condition, table-row reduce -- the decomposition of wav2letter_amd/csrc/attention_fused_bwd.hip) against
torch float64 autograd of the attention forward with the reference's relative-position rotation (oracle/transformer_oracle.py;
recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151).  No rounding anywhere: agreement to 1e-12."""
import numpy as np
import pytest
import torch

from oracle import attention_bwd_model as M
from oracle import transformer_oracle as TO


@pytest.mark.parametrize("B,H,T,d,csz,ragged", [(1, 2, 64, 8, 7, False), (2, 1, 37, 8, 460, False), (1, 1, 70, 4, 40, True),
                                                 (1, 1, 130, 4, 460, False), (2, 2, 33, 4, 0, True), (3, 1, 1, 4, 2, False),
                                                 (1, 1, 96, 4, 3, False), (1, 1, 190, 4, 100, True), (1, 1, 32, 4, 32, False),
                                                 (1, 1, 65, 4, 65, False)])
def test_fused_backward_decomposition_is_the_gradient(B, H, T, d, csz, ragged):
    g = torch.Generator().manual_seed(100 * T + csz)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    q, k, v, dctx = (mk(B, H, T, d).requires_grad_(True) for _ in range(4))
    E = (mk(max(1, 2 * csz - 1), d) * 0.5).requires_grad_(True)
    scale = 1.0 / np.sqrt(d)
    S = q @ k.transpose(-1, -2)
    if csz:
        rot = TO.relative_position_rotate(q @ E.t())
        n = E.shape[0] // 2
        S = S + rot[..., n:n + T]
    S = S * scale
    if ragged:
        keyLen = torch.tensor([T] + [max(1, (T * (3 + b)) // (5 + b)) for b in range(1, B)])
        if B == 1:
            keyLen = torch.tensor([max(1, (2 * T) // 3)])
        S = S.masked_fill(torch.arange(T)[None, None, None, :] >= keyLen[:, None, None, None], float("-inf"))
    P = torch.softmax(S, dim=-1)
    ((P @ v) * dctx.detach()).sum().backward()
    n_ = lambda x: x.detach().numpy()
    dq, dk, dv, dE = M.fused_backward_model(n_(q), n_(k), n_(v), n_(E) if csz else None, n_(P), n_(dctx), scale, csz)
    tol = lambda ref: 1e-12 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(dq - n_(q.grad)).max() < tol(n_(q.grad))
    assert np.abs(dk - n_(k.grad)).max() < tol(n_(k.grad))
    assert np.abs(dv - n_(v.grad)).max() < tol(n_(v.grad))
    if csz:
        assert np.abs(dE - n_(E.grad)).max() < tol(n_(E.grad))
        # rows outside the window an utterance of T frames reaches stay zero
        n0 = csz - 1
        reach = np.zeros(2 * csz - 1, bool)
        reach[max(0, n0 - (T - 1)):min(2 * csz - 1, n0 + T)] = True
        assert np.abs(n_(E.grad)[~reach]).max(initial=0.0) == 0.0 and np.abs(dE[~reach]).max(initial=0.0) == 0.0


def test_tile_count_and_c_layout():
    assert [M.nt_of(t) for t in (1, 64, 65, 128, 129, 192)] == [2, 2, 4, 4, 6, 6]
    with pytest.raises(ValueError):
        M.nt_of(193)
    rows = np.concatenate([M.c_layout_rows(0), M.c_layout_rows(1)])
    assert sorted(rows.tolist()) == list(range(32))
