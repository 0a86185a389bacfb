"""The CPU oracle must reproduce the reference's outputs bit for bit on every golden case.

Fixtures: ``tests/golden/cleverhans_golden.npz`` (made by ``tests/golden/make_golden.py`` from the
reference's own functions).  Single-threaded so that reduction order matches the generator's.
"""
import numpy as np
import pytest
import torch

from tests.adapters import OracleImpl
from tests.golden.cases import ALL_CASES, run_case


@pytest.fixture(scope="module", autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
def test_oracle_matches_reference_bitwise(case, golden):
    res = run_case(OracleImpl(), case, "cpu")
    for key, val in res.items():
        want = golden["{}/{}".format(case["name"], key)]
        got = val.detach().cpu().numpy()
        assert got.shape == want.shape, key
        assert np.array_equal(got, want, equal_nan=True), (case["name"], key, np.abs(got - want).max())


def test_golden_covers_every_case(golden):
    names = {k.split("/")[0] for k in golden}
    assert names == {c["name"] for c in ALL_CASES}
