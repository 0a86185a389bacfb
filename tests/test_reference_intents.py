"""Intent of the reference's own attack tests, restated for the operators it actually ships.

``cleverhans/torch/tests/test_attacks.py`` (``TestProjectedGradientMethod`` :197-545) is stale in the reference tree -- it
imports modules that are absent and calls the pre-modification signatures -- so it pins nothing
executable.  What it SPECIFIES is restated here against the shipped path with the toy white box: argument validation
(also in tests/test_host_logic.py), clip bounds (test_clips :496-521), the eps-ball (test_eps/test_clip_eta), few small
steps stay far from the ball's boundary (test_do_not_reach_lp_boundary :428-458), the attack makes progress
(test_attack_strength :460-476) and random restarts differ (test_multiple_initial_random_step :527-545).
"""
import numpy as np
import pytest
import torch

from tests.toy_models import ToyWhiteBox, toy_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def setup():
    import vqattack_amd
    toy = ToyWhiteBox(device=DEV)
    x0, _ = toy_inputs(4)
    x0 = x0.to(DEV)
    g = torch.Generator().manual_seed(3)
    start = torch.clamp(x0 + torch.empty(x0.shape).uniform_(-0.05, 0.05, generator=g).to(DEV), -1, 1)
    with torch.no_grad():
        y = toy.albef_feats(x0)
    pgd = vqattack_amd.dropin.load("albef").projected_gradient_descent.projected_gradient_descent
    return toy, x0, start, y, pgd


@pytest.mark.parametrize("norm", [np.inf, 2])
def test_clips_and_eps_ball(setup, norm):
    toy, x0, start, y, pgd = setup
    eps, eps_iter = (0.3, 0.03) if norm == np.inf else (3.0, 0.5)
    adv, losses = pgd(toy.albef_feats, start, eps, eps_iter, 30, norm, clip_min=-1.0, clip_max=1.0, y=list(y), ori_x=x0,
                      time=1, ls=1)
    assert torch.all(adv <= 1.0) and torch.all(adv >= -1.0)
    delta = adv - x0
    if norm == np.inf:
        assert float(delta.abs().max()) <= np.float32(eps) + 1e-7
    else:
        assert float(delta.flatten(1).norm(dim=1).max()) <= eps * (1 + 1e-5)
    assert len(losses) == 30 and losses[-1] > losses[0]          # the attack maximises the feature loss


@pytest.mark.parametrize("norm", [np.inf, 2])
def test_do_not_reach_lp_boundary(setup, norm):
    toy, x0, _, y, pgd = setup
    g = torch.Generator().manual_seed(4)
    start = torch.clamp(x0 + torch.empty(x0.shape).uniform_(-1e-3, 1e-3, generator=g).to(DEV), -1, 1)
    adv, _ = pgd(toy.albef_feats, start, 0.5 if norm == np.inf else 40.0, 0.01, 10, norm, y=list(y), ori_x=x0,
                 time=1, ls=1)
    d = adv - x0
    delta = d.abs().flatten(1).max(dim=1).values if norm == np.inf else d.flatten(1).norm(dim=1)
    assert float(delta.max()) <= 10 * 0.01 + 2e-3 * (1 if norm == np.inf else 56)   # 10 steps of size 0.01 at most


def test_no_clip_and_one_sided_clip(setup):
    toy, x0, start, y, pgd = setup
    adv, _ = pgd(toy.albef_feats, start, 0.2, 0.05, 6, np.inf, y=list(y), ori_x=x0, time=1, ls=1)
    assert float((adv - x0).abs().max()) <= np.float32(0.2) + 1e-7
    assert float(adv.abs().max()) > 1.0                           # nothing clamps to the image range
    with pytest.raises(ValueError):                               # the FGM step rejects one-sided clipping
        pgd(toy.albef_feats, start, 0.2, 0.05, 2, np.inf, clip_max=1.0, y=list(y), ori_x=x0, time=1, ls=1,
            sanity_checks=False)


def test_random_restarts_differ_and_stay_feasible(setup):
    toy, x0, _, y, pgd = setup
    torch.manual_seed(0)
    outs = [pgd(toy.albef_feats, x0, 0.125, 0.01, 2, np.inf, clip_min=-1, clip_max=1, y=list(y), ori_x=x0, time=0,
                ls=1)[0] for _ in range(3)]
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    for o in outs:
        assert float((o - x0).abs().max()) <= np.float32(0.125) + 1e-7
    same = pgd(toy.albef_feats, x0, 0.125, 0.01, 2, np.inf, clip_min=-1, clip_max=1, y=list(y), ori_x=x0, time=0, ls=1,
               rand_minmax=0.0)[0]                                 # a zero-width draw is the deterministic start
    again = pgd(toy.albef_feats, x0, 0.125, 0.01, 2, np.inf, clip_min=-1, clip_max=1, y=list(y), ori_x=x0, time=0,
                ls=1, rand_minmax=0.0)[0]
    assert torch.equal(same, again)


def test_targeted_descends(setup):
    toy, x0, start, y, pgd = setup
    _, up = pgd(toy.albef_feats, start, 0.125, 0.01, 8, np.inf, clip_min=-1, clip_max=1, y=list(y), ori_x=x0, time=1,
                ls=1)
    _, down = pgd(toy.albef_feats, start, 0.125, 0.01, 8, np.inf, clip_min=-1, clip_max=1, y=list(y), ori_x=x0, time=1,
                  ls=1, targeted=True)
    assert up[-1] > up[0]
    assert down[-1] > down[0] and down[0] == pytest.approx(-up[0], rel=1e-5)   # minimises the loss: reports -loss rising
