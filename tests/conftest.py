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


def usable_cores(cap=16):
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota.  A one-GPU job on the GPU box
    gets a 16-core share of a much larger host, and torch sizes its intra-op pool by the HOST's core count: the CPU
    oracle's many small ops then wake a hundred threads that share 16 cores."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


@pytest.fixture(scope="session", autouse=True)
def _cpu_thread_pool():
    """Size torch's CPU thread pool for the cores the job really has (the CPU oracle side of every parity test)."""
    import torch
    before = torch.get_num_threads()
    torch.set_num_threads(usable_cores())
    yield
    torch.set_num_threads(before)


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
