"""Randomised range check of FCC / FAC / CTC against the fp64 oracle (tools/exp/criterion_fuzz.py): label-set sizes on both sides of
every kernel switch (N <= 31, <= 64, large), lattices from 1 to 300 positions, T from 1 to 2000, every scale mode, emission and
transition magnitudes up to 20 / 8 nats of sigma -- several times what a recipe produces.  The envelope is deliberate: with emissions
AND transitions tens of nats wide (sigma 50 x sigma >= 20) the FullConnectionCriterion kernels for MORE THAN 31 labels -- scaled
exp-domain recursions in fp32 -- clamp a state that is more than ~87 nats behind the frame's best, which can move the posterior to another path (loss still at
1e-4, gradients not; DESIGN 1, profiles/r05_run32_criterion_fuzz.log); the reference's log-domain recursion has no such limit."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("seed", [11, 12])
def test_criteria_random_shapes_and_magnitudes(seed):
    from tools.exp.criterion_fuzz import run
    bad = run(40, seed, x_scales=(0.1, 1.0, 5.0, 20.0), a_scales=(0.0, 0.3, 2.0, 8.0), verbose=False)
    assert not bad, "\n".join(bad)


def test_full_connection_up_to_31_labels_with_very_wide_dynamics():
    """N <= 31 (the letter-based ASG recipes): the utterances the linear-domain scans flag re-run on a true log-domain pair
    (fcc_fwd_log / fcc_bwd_log: every term one exponential of a non-positive sum) -- exact where the scaled-exp kernels clamp"""
    from tools.exp.criterion_fuzz import run
    import re
    bad = run(30, 9, x_scales=(5.0, 20.0, 50.0), a_scales=(8.0, 20.0, 40.0), verbose=False, n_choices=(3, 5, 16, 29, 30, 31))
    fcc_bad = [l for l in bad if re.search(r"FCC [^ ]+ BAD", l)]
    assert not fcc_bad, "\n".join(fcc_bad)


def test_force_alignment_with_very_wide_transitions():
    """FAC alone stays exact far outside that envelope (fp64 mantissas, integer exponents per position; kappa of any range)"""
    from tools.exp.criterion_fuzz import run
    import re
    bad = run(30, 7, x_scales=(1.0, 20.0, 50.0), a_scales=(20.0, 40.0), verbose=False)
    fac_bad = [l for l in bad if re.search(r"FAC [^ ]+ BAD", l)]
    assert not fac_bad, "\n".join(fac_bad)
