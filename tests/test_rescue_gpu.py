"""GPU parity test: the rescue-scan kernel (bbpipe_quick_rescue_device) against the CPU oracle."""
import pytest

from bbmap_amd.rescue import quick_rescue_batch
from oracle.oracle import quick_rescue
from tests.rescue_problems import make_problems

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("affine", [True, False])
def test_quick_rescue_matches_oracle(affine):
    ref, probs = make_problems(11, 1500)
    got = quick_rescue_batch(probs, [ref], min_index=[300], use_affine=affine)
    found = 0
    for p, g in zip(probs, got):
        exp = quick_rescue(p[0], ref, 300, p[2], p[3], p[4], p[5], p[6], useAffine=affine)
        assert g == exp, (p[1:], g, exp)
        found += exp is not None
    assert found > 600
