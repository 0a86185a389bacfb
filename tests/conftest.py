import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _kernel_library():
    """Build the HIP kernel library when it is missing or older than its sources (a fresh clone has no .so: built
    artefacts are git-ignored).  hipcc cross-compiles gfx950 without a GPU; a failure here fails the run loudly."""
    from vqattack_amd import build
    build.build(force=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "cleverhans_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
