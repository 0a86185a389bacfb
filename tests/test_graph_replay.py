"""hipGraph replay of PGD iterations must reproduce the eager loop bit for bit (same kernels, same order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("batch", [1, 3])
def test_graphed_pgd_equals_eager(batch):
    import vqattack_amd
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
    cfg = vlmo_tiny()
    model = FrozenVlmo(cfg, seed=5).to(DEV)
    ad = VlmoAttackAdapters(model)
    ids = torch.tensor([[101, 2054, 3609, 2003, 102, 0, 0, 0]] * batch, device=DEV)
    ad.set_text(ids, (ids != 0).long())
    g = torch.Generator().manual_seed(2)
    x0 = torch.empty(batch, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g).to(DEV)
    eta = torch.empty(x0.shape).uniform_(-0.125, 0.125, generator=g).to(DEV)
    y = ad.gen_ori_feats(x0)
    kw = dict(clip_min=-1, clip_max=1, ori_x=x0, time=0, ls=1, flavor="vlmo", init_eta=eta)
    with torch.enable_grad():
        adv_e, loss_e = vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 7, np.inf, y=list(y), **kw)
        adv_g, loss_g = vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 7, np.inf, y=list(y),
                                                                graph=True, **kw)
    assert torch.equal(adv_e, adv_g)
    assert loss_e == loss_g and len(loss_g) == 7
    with pytest.raises(ValueError):
        vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 2.0, 0.5, 2, 2, y=list(y), graph=True,
                                                clip_min=-1, clip_max=1, ori_x=x0, time=1, ls=1, flavor="vlmo")


def test_environment_opt_in_replays_eligible_calls_only(monkeypatch):
    """``VQA_PGD_GRAPH=1``: an unmodified driver's L-inf feature-loss call is replayed (same bits as eager), its L2 call
    is left alone instead of raising."""
    import vqattack_amd
    from vqattack_amd import attacks
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
    cfg = vlmo_tiny()
    model = FrozenVlmo(cfg, seed=5).to(DEV)
    ad = VlmoAttackAdapters(model)
    ids = torch.tensor([[101, 2054, 3609, 2003, 102, 0, 0, 0]], device=DEV)
    ad.set_text(ids, (ids != 0).long())
    g = torch.Generator().manual_seed(3)
    x0 = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g).to(DEV)
    eta = torch.empty(x0.shape).uniform_(-0.125, 0.125, generator=g).to(DEV)
    y = ad.gen_ori_feats(x0)
    kw = dict(clip_min=-1, clip_max=1, ori_x=x0, time=0, ls=1, flavor="vlmo", init_eta=eta)
    calls = []
    real = attacks._graphed_linf_loop
    monkeypatch.setattr(attacks, "_graphed_linf_loop", lambda *a, **k: calls.append(1) or real(*a, **k))
    with torch.enable_grad():
        adv_e, loss_e = vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 5, np.inf, y=list(y), **kw)
        assert not calls
        monkeypatch.setenv("VQA_PGD_GRAPH", "1")
        adv_g, loss_g = vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 5, np.inf, y=list(y), **kw)
        assert calls == [1]
        vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 5, np.inf, y=list(y), graph=False, **kw)
        vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 2.0, 0.5, 2, 2, y=list(y), clip_min=-1, clip_max=1,
                                                ori_x=x0, time=1, ls=1, flavor="vlmo")
        assert calls == [1]
    assert torch.equal(adv_e, adv_g) and loss_e == loss_g


def test_runner_with_graph_replay_equals_eager():
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
    cfg = vlmo_tiny()
    model = FrozenVlmo(cfg, seed=6).to(DEV)
    ids = torch.tensor([[101, 2054, 3609, 2003, 102, 0, 0, 0], [101, 2129, 2116, 6077, 102, 0, 0, 0]], device=DEV)
    masks = (ids != 0).long()
    g = torch.Generator().manual_seed(8)
    img = torch.empty(2, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g).to(DEV)
    eta = torch.empty(img.shape).uniform_(-0.125, 0.125, generator=g).to(DEV)
    none = torch.zeros_like(ids, dtype=torch.bool)
    outs = []
    for use_graph in (False, True):
        atk = BatchedVQAttack(VlmoAttackAdapters(model), "vlmo", model.embedding_tables(),
                              AttackConfig(budget=9, use_graph=use_graph))
        outs.append(atk.attack_batch(img, ids, masks, none, init_eta=eta))
    assert torch.equal(outs[0].adv_images, outs[1].adv_images)
    assert outs[0].loss_lists == outs[1].loss_lists


def test_patch_layout_state_equals_image_layout():
    """The PGD state kept patch-major (layout.py) gives bit-identical adversarial images."""
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
    cfg = vlmo_tiny()
    model = FrozenVlmo(cfg, seed=6).to(DEV)
    ids = torch.tensor([[101, 2054, 3609, 2003, 102, 0, 0, 0], [101, 2129, 2116, 6077, 102, 0, 0, 0]], device=DEV)
    masks = (ids != 0).long()
    g = torch.Generator().manual_seed(9)
    img = torch.empty(2, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g).to(DEV)
    eta = torch.empty(img.shape).uniform_(-0.125, 0.125, generator=g).to(DEV)
    none = torch.zeros_like(ids, dtype=torch.bool)
    outs = []
    for patch_layout in (False, True):
        atk = BatchedVQAttack(VlmoAttackAdapters(model), "vlmo", model.embedding_tables(),
                              AttackConfig(budget=7, patch_layout=patch_layout, sanity_checks=True))
        outs.append(atk.attack_batch(img, ids, masks, none, init_eta=eta))
    assert outs[1].adv_images.shape == img.shape
    assert torch.equal(outs[0].adv_images, outs[1].adv_images)
