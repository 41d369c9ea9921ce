"""Checkpoint container (reference parameter order / layouts) and learning-rate schedule -- host logic, no GPU."""
import math

import numpy as np
import pytest


def _trainer(crit="ctc", nlabel=12):
    from wav2letter_amd import recipes
    from wav2letter_amd.trainer import Trainer
    arch = recipes.tds_ctc_small_arch(c=(4, 6), h=8, kw=5)
    tr = Trainer(arch, 8, nlabel, crit, 4, 4.0 if crit == "asg" else 0.0, device="cpu")
    tr.init_params(seed=11)
    return arch, tr


@pytest.mark.parametrize("crit", ["ctc", "asg"])
def test_round_trip_in_reference_layout(tmp_path, crit):
    from wav2letter_amd import checkpoint
    arch, tr = _trainer(crit)
    path = str(tmp_path / "m.w2l")
    checkpoint.save(path, tr, arch, crit, step=123, flags={"lr": 0.1})
    header, arrays = checkpoint.read(path)
    assert header["step"] == 123 and header["criterion"] == crit and header["flags"] == {"lr": 0.1}
    table = tr.param_table()
    net = [t for t in header["tensors"] if t["kind"] == "network"]
    assert [t["name"] for t in net] == [r[0] for r in table]
    if crit == "asg":
        t = [t for t in header["tensors"] if t["kind"] == "criterion"][0]
        assert t["shape"] == [12, 12]
    # a fresh trainer with different weights is restored exactly
    arch2, tr2 = _trainer(crit)
    tr2.init_params(seed=99)
    assert not np.array_equal(tr2.host_params, tr.host_params)
    assert checkpoint.load(path, tr2, arch2) == 123
    assert np.array_equal(tr2.host_params, tr.host_params)


def test_mismatch_is_rejected(tmp_path):
    from wav2letter_amd import checkpoint
    arch, tr = _trainer("ctc")
    path = str(tmp_path / "m.w2l")
    checkpoint.save(path, tr, arch, "ctc")
    _, other = _trainer("ctc", nlabel=13)
    with pytest.raises(ValueError):
        checkpoint.load(path, other)
    with pytest.raises(ValueError):
        checkpoint.load(path, tr, arch + "\n# changed\nL 1 1\n")
    open(str(tmp_path / "bad"), "wb").write(b"not a checkpoint")
    with pytest.raises(ValueError):
        checkpoint.read(str(tmp_path / "bad"))
    raw = open(path, "rb").read()
    open(str(tmp_path / "cut"), "wb").write(raw[:len(raw) // 2])
    with pytest.raises(ValueError):
        checkpoint.read(str(tmp_path / "cut"))


def test_learning_rate_schedule_matches_the_reference_formula():
    from wav2letter_amd.checkpoint import learning_rate
    f = {"lr": 0.4, "warmup": 100, "gamma": 0.5, "stepsize": 1000, "lr_decay": 3, "lr_decay_step": 2}
    assert learning_rate(f, 0, 1) == 0.0
    assert learning_rate(f, 50, 1) == pytest.approx(0.4 * 0.5 ** 0.05 * 0.5)
    assert learning_rate(f, 1000, 2) == pytest.approx(0.4 * 0.5)
    assert learning_rate(f, 1000, 3) == pytest.approx(0.4 * 0.5 * 0.5)       # first decay epoch
    assert learning_rate(f, 1000, 5) == pytest.approx(0.4 * 0.5 * 0.25)      # 1 + (5-3)//2 halvings
    c = {"lr": 1.0, "lrcosine": True, "warmup": 1}
    assert learning_rate(c, 500, 1, n_batches=1000) == pytest.approx(math.cos(math.pi / 4))
    assert learning_rate(f, 200, 1, base=0.02) == pytest.approx(0.02 * 0.5 ** 0.2)


def test_criterion_kind_and_odd_transitions(tmp_path):
    """a CTC checkpoint is rejected by an ASG trainer (and the reverse); odd N: the transitions are saved as exactly
    (N, N) although the arena slot is padded"""
    from wav2letter_amd import checkpoint
    arch, tr = _trainer("asg", nlabel=11)
    path = str(tmp_path / "a.w2l")
    checkpoint.save(path, tr, arch, "asg")
    header, arrays = checkpoint.read(path)
    t = [t for t in header["tensors"] if t["kind"] == "criterion"][0]
    assert t["shape"] == [11, 11] and t["numel"] == 121
    arch2, tr2 = _trainer("asg", nlabel=11)
    tr2.init_params(seed=5)
    checkpoint.load(path, tr2, arch2)
    assert np.array_equal(tr2.host_params[:tr2.n_net + 121], tr.host_params[:tr.n_net + 121])
    _, trc = _trainer("ctc", nlabel=11)
    with pytest.raises(ValueError):
        checkpoint.load(path, trc, arch)
    pathc = str(tmp_path / "c.w2l")
    checkpoint.save(pathc, trc, arch, "ctc")
    with pytest.raises(ValueError):
        checkpoint.load(pathc, tr2, arch)
