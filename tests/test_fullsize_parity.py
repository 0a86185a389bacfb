"""Parity at the three white-box SHAPES of BASELINE.json's configs, on the MI355X, against the reference-structured CPU
oracle (``oracle/`` with reference-style batch-1 feature packing) on the same seeded inputs:

  * configs[1] / [3]: VLMO-base (12 x 768, 13 feature maps), 384 px, 40-token question, the FULL 40-step attack;
  * configs[2]:       ALBEF-base (ViT-B/16 + 12-layer BERT, fusion layer 6, 13 + 13 maps), 384 px, 8 steps
                      (``mlm_probability = 0``: the per-forward random token masking is switched off for determinism);
  * configs[4]:       VLMO-large (24 x 1024, 25 feature maps through ONE ``vqa_neg_cos_rows_multi`` launch), joint
                      image + text attack with 2 substitutable words on a short budget (6 image steps + 2 probe steps).

Stated fp32 tolerance for full-size perturbations: the white-box forward/backward runs in
PyTorch-ROCm on the GPU and in PyTorch on the CPU, so gradients differ in the last bits; with a sign step a pixel whose
gradient is ~0 can flip by 2 * eps_iter and the trajectories then drift apart slowly.  Required after a complete
40-step attack: >= 99 % of the pixels bit-identical, no pixel further than 4 * eps_iter from the reference, mean
|deviation| <= 1e-4; after <= 8 steps: >= 99.9 % bit-identical, max deviation 2 * eps_iter * steps.  Loss trajectories:
1e-4 relative.  Substituted token ids: equal.
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
EPS, EPS_ITER = 0.125, 0.01


def _cpu_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


@pytest.fixture(scope="module", autouse=True)
def _threads():
    before = torch.get_num_threads()
    torch.set_num_threads(_cpu_threads())
    yield
    torch.set_num_threads(before)


def _inputs(n_tokens_list, text_len, seed=0):
    ids = torch.zeros(len(n_tokens_list), text_len, dtype=torch.long)
    g = torch.Generator().manual_seed(seed)
    for s, n in enumerate(n_tokens_list):
        ids[s, 0] = 101
        ids[s, 1:1 + n] = torch.randint(1000, 30522, (n,), generator=g)
        ids[s, 1 + n] = 102
    img = torch.empty(len(n_tokens_list), 3, 384, 384).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(img).uniform_(-EPS, EPS, generator=g)
    return ids, (ids != 0).long(), img, eta


def _compare(got, want, steps, full_attack):
    same = float((got == want).float().mean())
    dev = (got - want).abs()
    if full_attack:
        assert same >= 0.99, "only {:.3%} of the pixels are bit-identical".format(same)
        assert float(dev.max()) <= 4 * EPS_ITER + 1e-6
        assert float(dev.mean()) <= 1e-4
    else:
        assert same >= 0.999, "only {:.3%} of the pixels are bit-identical".format(same)
        assert float(dev.max()) <= 2 * EPS_ITER * steps + 1e-6
    return same


def test_vlmo_base_full_40_step_attack_matches_cpu_oracle():
    """BASELINE configs[1]/[3] shape, the complete attack budget."""
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    steps = 40
    cpu_model = FrozenVlmo(vlmo_base(384), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([5], 40)
    attack = BatchedVQAttack(VlmoAttackAdapters(gpu_model), "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=steps, sanity_checks=True))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV))
    ad = VlmoRefAdapters(cpu_model, ids, masks)
    with torch.enable_grad():
        adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img, EPS, EPS_ITER, steps, np.inf, clip_min=-1,
                                                        clip_max=1, y=ad.gen_ori_feats(img), ori_x=img, time=0, ls=1,
                                                        flavor="vlmo", init_eta=eta)
    _compare(res.adv_images[0].cpu(), adv[0].detach(), steps, full_attack=True)
    assert len(res.loss_lists[0]) == steps
    np.testing.assert_allclose(res.loss_lists[0], losses, rtol=1e-4)


def test_albef_base_attack_matches_cpu_oracle():
    """BASELINE configs[2] shape (ALBEF: [text maps, image maps], 13 + 13 per-layer tensors, cross-attention fusion)."""
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import AlbefRefAdapters
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base
    steps = 8
    cpu_model = FrozenAlbef(albef_base(384, mlm_probability=0.0), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([6], 8)             # ALBEF pads to the longest question: 8 tokens, no padding
    attack = BatchedVQAttack(AlbefAttackAdapters(gpu_model), "albef", gpu_model.embedding_tables(),
                             AttackConfig(budget=steps, sanity_checks=True))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV))
    ad = AlbefRefAdapters(cpu_model, ids, masks)
    tgt = ad.gen_ori_feats(img)
    with torch.enable_grad():
        adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img, EPS, EPS_ITER, steps, np.inf, clip_min=-1,
                                                        clip_max=1, y=[tgt[0], tgt[1], None, None, None], ori_x=img,
                                                        time=0, ls=1, flavor="albef", init_eta=eta)
    _compare(res.adv_images[0].cpu(), adv[0].detach(), steps, full_attack=False)
    np.testing.assert_allclose(res.loss_lists[0], losses, rtol=1e-4)


def test_vlmo_large_joint_attack_matches_cpu_oracle():
    """BASELINE configs[4] shape: VLMO-large, joint image + text attack (25 maps per loss launch, D = 1024, text-gradient
    probes, candidate scoring, acceptance and masked-token embedding substitution on the device)."""
    from oracle import attack_loop
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_large
    budget, words = 6, 2                               # blocks [2, 2, 2] + 2 probe steps = 8 white-box gradient steps
    cpu_model = FrozenVlmo(vlmo_large(384), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    assert len(gpu_model.blocks) == 24 and gpu_model.cfg.dim == 1024
    ids, masks, img, eta = _inputs([6], 40)
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[:, 1:1 + words] = True
    adapters = VlmoAttackAdapters(gpu_model)
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att, threshold=0)
    assert len(proposals[0]) == words
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters, "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == budget + words
    adv, new_ids, losses = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img, ids, masks, proposals[0], sim,
                                                  init_eta=eta, budget=budget, sim_threshold=0.3)
    assert res.adv_text_ids[0].cpu().tolist() == new_ids[0].tolist()
    assert int((new_ids[0] != ids[0]).sum()) >= 1, "the case should exercise at least one accepted substitution"
    _compare(res.adv_images[0].cpu(), adv[0].detach(), budget + words, full_attack=False)
    for got, want in zip(res.loss_lists, losses):
        np.testing.assert_allclose(got, want, rtol=1e-4)


def test_vlmo_base_attack_is_bitwise_reproducible():
    """Two runs of the same attack give the same bits: no kernel of the path accumulates with float atomics (loss fold
    in index order, two-stage per-sample reductions, attention backward without atomics), so a result can be compared
    across runs, ranks and batch compositions."""
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    steps = 6
    model = FrozenVlmo(vlmo_base(384), seed=0).to(DEV)
    ids, masks, img, eta = _inputs([5, 9, 7], 40)
    runs = []
    for _ in range(2):
        attack = BatchedVQAttack(VlmoAttackAdapters(model), "vlmo", model.embedding_tables(),
                                 AttackConfig(budget=steps, sanity_checks=True))
        res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV),
                                  torch.zeros_like(ids, dtype=torch.bool).to(DEV), init_eta=eta.to(DEV))
        runs.append((res.adv_images.clone(), [list(x) for x in res.loss_lists]))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]
