"""The plain-C restatement of the L-inf iteration tail agrees bit for bit with the golden-pinned torch oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cleverhans_cpu as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_linf.so")


@pytest.fixture(scope="module")
def clib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(LIB)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_linf_step.argtypes = [fp, fp, fp, fp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [ctypes.c_int]
    lib.oracle_range_ok.argtypes = [fp, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    lib.oracle_range_ok.restype = ctypes.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


@pytest.mark.parametrize("clip", [True, False])
@pytest.mark.parametrize("eps,eps_iter", [(0.125, 0.01), (8 / 255, 2 / 255)])
def test_c_restatement_equals_torch_oracle(clib, clip, eps, eps_iter):
    r = np.random.RandomState(5)
    n = 3 * 64 * 64 + 3
    x0 = r.uniform(-1, 1, n).astype(np.float32)
    x = np.clip(x0 + r.uniform(-eps, eps, n).astype(np.float32), -1, 1)
    g = r.standard_normal(n).astype(np.float32)
    g[::7] = 0.0
    g[1::11] = -0.0
    g[2::13] = np.nan
    g[3::17] = np.float32(1e-42)
    x[5] = np.nan
    x0[9] = np.nan
    out = np.empty_like(x)
    clib.oracle_linf_step(_p(x), _p(g), _p(x0), _p(out), n, eps_iter, eps, -1.0, 1.0, int(clip))
    want = o.pgd_tail_given_grad(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(x0), eps_iter, eps, np.inf,
                                 *((-1, 1) if clip else (None, None))).numpy()
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_c_range_flag(clib):
    x = np.linspace(-1, 1, 101).astype(np.float32)
    assert clib.oracle_range_ok(_p(x), x.size, -1.0, 1.0) == 1 and o.range_ok(torch.from_numpy(x), -1, 1)
    for bad in (1.0001, -3.0, np.nan):
        y = x.copy()
        y[50] = bad
        assert clib.oracle_range_ok(_p(y), y.size, -1.0, 1.0) == 0 and not o.range_ok(torch.from_numpy(y), -1, 1)
