"""CPU-side checks of the shipped host code: argument validation happens before any device work, CPU tensors are
refused loudly (no fallback), and the drop-in packages expose the reference's module paths."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import vqattack_amd
from vqattack_amd._hip import HipExtensionError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X = torch.zeros(1, 3, 4, 4)


def _fn(t):
    return [t, t]


@pytest.mark.parametrize("flavor", ["albef", "vlmo"])
def test_validation_precedes_device_checks(flavor):
    from vqattack_amd import dropin
    pgd = dropin.load(flavor).projected_gradient_descent.projected_gradient_descent
    fgm = dropin.load(flavor).fast_gradient_method.fast_gradient_method
    with pytest.raises(NotImplementedError):
        pgd(_fn, X, 0.1, 0.01, 1, 1, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, 0.01, 1, 7, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, -1, 0.01, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, -1, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(AssertionError):
        pgd(_fn, X, 0.1, 0.5, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, 0.01, 1, np.inf, clip_min=2, clip_max=1, ori_x=X, ls=1)
    assert pgd(_fn, X, 0, 0.01, 1, np.inf, ori_x=X, ls=1) is X
    assert pgd(_fn, X, 0.1, 0, 1, np.inf, ori_x=X, ls=1) is X
    with pytest.raises(ValueError):
        fgm(_fn, X, 0.1, 5, X, ls=1)
    with pytest.raises(ValueError):
        fgm(_fn, X, -0.1, np.inf, X, ls=1)
    assert fgm(_fn, X, 0, np.inf, X, ls=1) is X


def test_cpu_tensors_are_refused_not_emulated():
    with pytest.raises(HipExtensionError, match="no CPU fallback"):
        vqattack_amd.projected_gradient_descent(_fn, X, 0.1, 0.01, 1, np.inf, ori_x=X, ls=1, y=[X, X])
    with pytest.raises(HipExtensionError):
        vqattack_amd.fast_gradient_method(_fn, X, 0.1, np.inf, X, ls=1, y=[X, X])
    with pytest.raises(HipExtensionError):
        vqattack_amd.clip_eta(X, np.inf, 0.1)
    with pytest.raises(HipExtensionError):
        vqattack_amd.optimize_linear(X, 0.1, 2)
    with pytest.raises(NotImplementedError):        # reference behaviour, raised before the device check
        vqattack_amd.clip_eta(X, 1, 0.1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vqattack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, f)


@pytest.mark.parametrize("flavor", ["albef", "vlmo"])
def test_dropin_module_paths(flavor):
    root = os.path.join(ROOT, "vqattack_amd", "dropin", flavor)
    for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
        del sys.modules[name]
    sys.path.insert(0, root)
    try:
        pgd = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent")
        pgd_vl = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent_vl")
        fgm = importlib.import_module("cleverhans.torch.attacks.fast_gradient_method")
        fgm_vl = importlib.import_module("cleverhans.torch.attacks.fast_gradient_method_vl")
        utils = importlib.import_module("cleverhans.torch.utils")
        assert pgd.projected_gradient_descent.keywords == {"flavor": flavor}
        assert pgd_vl.projected_gradient_descent.keywords == {"flavor": flavor}
        assert fgm.fast_gradient_method.keywords == {"flavor": flavor}
        assert fgm_vl.fast_gradient_method.keywords == {"flavor": flavor}
        assert callable(utils.clip_eta) and callable(utils.optimize_linear)
    finally:
        sys.path.remove(root)
        for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
            del sys.modules[name]
