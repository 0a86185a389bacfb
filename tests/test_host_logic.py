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


def test_attention_bias_layout_decision():
    """Which bias tensors the attention kernels may read in place (rows readable to ceil32(Sk), aligned strides) and
    which must be copied into padded rows first -- the rule of `attention._bias_layout`, on host tensors."""
    import torch
    from vqattack_amd import attention

    s = 587
    slab = torch.zeros(1, 12, s, 608)                       # what FrozenVlmo.attention_bias builds: rows padded to 32
    view, ok = attention._bias_layout(slab[..., :s].expand(64, -1, -1, -1), s)
    assert ok and view.shape == (1, 12, s, s) and view.data_ptr() == slab.data_ptr()
    _, ok = attention._bias_layout(torch.zeros(1, 12, s, s), s)          # dense rows of 587 floats: no room, odd pitch
    assert not ok
    _, ok = attention._bias_layout(torch.zeros(2, 3, 64, 64), 64)        # Sk a multiple of 32: dense rows are fine
    assert ok
    tight = torch.zeros(1, 1, 4, 96)[..., :70]                           # pitch 96 = ceil32(70): exactly enough room
    assert attention._bias_layout(tight, 70)[1]
    short = torch.zeros(1, 1, 4, 80)[..., :70]                           # last row ends 16 floats early
    assert not attention._bias_layout(short, 70)[1]
    pad_mask = torch.zeros(8, 1, 1, 40).expand(8, 12, 40, 40)            # key padding broadcast over heads and rows
    view, ok = attention._bias_layout(pad_mask, 40)
    assert view.shape == (8, 1, 1, 40) and not ok                        # 40-float rows: copied, but only 8 of them
    with pytest.raises(ValueError):
        attention._bias_layout(torch.zeros(1, 1, 4, 33), 40)
