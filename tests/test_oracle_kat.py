"""Known-answer tests of the reference for clip_eta / optimize_linear, restated for the oracle.

Source of the cases: ``VLMO_VQAttack/cleverhans/cleverhans/torch/tests/test_utils.py:23-111``
(identical in the ALBEF copy).  Same inputs, same acceptance criteria; run against ``oracle/``.
The GPU twin of this file is ``tests/test_hip_utils.py``.
"""
import numpy as np
import pytest
import torch

from oracle import cleverhans_cpu as o

EPS_LIST = [0, 0.1, 1.0, 3]


@pytest.fixture()
def rnd():
    g = torch.Generator().manual_seed(1234)
    return torch.randn(100, 3, 2, generator=g), torch.randn(100, 3, 2, generator=g)


def test_optimize_linear_linf():                       # test_utils.py:23-30
    grad = torch.tensor([[1.0, -2.0]])
    eta = o.optimize_linear(grad, eps=1.0, norm=np.inf)
    assert eta.shape == grad.shape
    assert np.allclose(torch.sum(grad * eta), grad.abs().sum())
    assert np.allclose(eta.abs(), 1.0)


def test_optimize_linear_l2():                         # :32-39
    grad = torch.tensor([[0.5 ** 0.5, -(0.5 ** 0.5)]])
    eta = o.optimize_linear(grad, eps=1.0, norm=2)
    assert np.allclose(torch.sum(grad * eta), 1.0)
    assert np.allclose(eta.pow(2).sum().sqrt(), 1.0)


@pytest.mark.parametrize("grad", [[[1.0, -2.0]], [[2.0, -2.0]]])   # :41-57 (incl. ties)
def test_optimize_linear_l1(grad):
    grad = torch.tensor(grad)
    eta = o.optimize_linear(grad, eps=1.0, norm=1)
    assert np.allclose(torch.sum(grad * eta), 2.0)
    assert np.allclose(eta.abs().sum(), 1.0)


@pytest.mark.parametrize("eps", EPS_LIST)
def test_norm_constraints(rnd, eps):                   # :59-91
    grad, _ = rnd
    red = [1, 2]
    assert np.allclose(o.optimize_linear(grad, eps=eps, norm=np.inf).abs(), eps)
    l1 = o.optimize_linear(grad, eps=eps, norm=1).abs().sum(dim=red)
    assert torch.allclose(l1, eps * torch.ones_like(l1))
    eta = o.optimize_linear(grad, eps=eps, norm=2)
    tiny = torch.tensor(1e-12)
    sq = torch.max(tiny, torch.sum(grad ** 2, red, keepdim=True))
    nrm = eta.pow(2).sum(dim=red, keepdim=True).sqrt()
    one = (sq <= tiny).to(torch.float) * nrm + (sq > tiny).to(torch.float)
    assert torch.allclose(nrm, eps * one)


def test_clip_eta(rnd):                                # :93-111
    _, eta = rnd
    c = o.clip_eta(eta.clone(), norm=np.inf, eps=0.5)
    assert torch.all(c <= 0.5) and torch.all(c >= -0.5)
    with pytest.raises(NotImplementedError):
        o.clip_eta(eta.clone(), norm=1, eps=0.5)
    c2 = o.clip_eta(eta.clone(), norm=2, eps=0.5)
    assert torch.all(c2.pow(2).sum(dim=[1, 2]).pow(0.5) <= 0.5001)
    with pytest.raises(ValueError):
        o.clip_eta(eta.clone(), norm=3, eps=0.5)
