"""Every parity case of tests/parity_cases.py (fixtures of the reference's eager path and Triton kernels, BASELINE.json
configs 1-4 at full size, the row-wise siblings): measured error <= enforced bound, tensor by tensor.  The same cases feed
tools/parity_report.py -> profiles/parity_rNN.json."""
import pytest

from parity_cases import ALL_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", ALL_CASES, ids=[n for n, _ in ALL_CASES])
def test_parity_case(name, fn):
    recs = fn()
    assert recs
    bad = [r for r in recs if not (r["err"] <= r["bound"])]
    assert not bad, bad
