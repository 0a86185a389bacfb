import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-rank GPU tests start their ranks through a fork server that is launched HERE, before anything in this
    # process touches the GPU: a process that has initialised HIP must never exec another program on the GPU boxes, and
    # the ranks must not inherit an initialised runtime either.  The server itself imports nothing GPU-related; every
    # rank is forked from it and initialises the GPU on its own.
    try:
        import multiprocessing
        from multiprocessing import forkserver
        multiprocessing.get_context("forkserver")
        forkserver.set_forkserver_preload([])
        forkserver.ensure_running()
    except (ImportError, ValueError, OSError):
        pass


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
