"""Parity at the reference's OWN resolution: 480 x 480 images (``ALBEF_attack/configs/VQA.yaml:10`` ``image_res: 480``;
``vlmo/config.py:283-299`` ``task_finetune_vqa_base_image480``) = 30 x 30 + 1 = 901 image tokens; VLMo runs
40 + 901 = 941 tokens (915 after the padding trim of a 12-word batch), ALBEF's fusion layers attend <= 25 text queries
over 901 image keys.  BASELINE.json's metric is quoted at 384 px, which every other full-size test uses; this file
is the same comparison -- the product on the MI355X against the CPU oracle with reference-style batch-1 packing -- at
the layout the reference's README runs (``entry/run.py with task_finetune_vqa_base_image480``).

Tolerances as tests/test_fullsize_parity.py for <= 8 steps: >= 99.9 % of the pixels bit-identical, |dev| <= 2 eps_iter
steps, losses 1e-4 relative, substituted ids equal.
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
EPS, EPS_ITER = 0.125, 0.01


@pytest.fixture(scope="module", autouse=True)
def _threads():
    from tests.test_fullsize_parity import _cpu_threads
    before = torch.get_num_threads()
    torch.set_num_threads(_cpu_threads())
    yield
    torch.set_num_threads(before)


def _inputs(n_words, text_len, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(len(n_words), text_len, dtype=torch.long)
    for s, n in enumerate(n_words):
        ids[s, 0] = 101
        ids[s, 1:1 + n] = torch.randint(1000, 30522, (n,), generator=g)
        ids[s, 1 + n] = 102
    img = torch.empty(len(n_words), 3, 480, 480).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(img).uniform_(-EPS, EPS, generator=g)
    return ids, (ids != 0).long(), img, eta


def _compare(got, want, steps):
    same = float((got == want).float().mean())
    assert same >= 0.999, "only {:.3%} of the pixels are bit-identical".format(same)
    assert float((got - want).abs().max()) <= 2 * EPS_ITER * steps + 1e-6
    return same


def test_vlmo_base_480px_ragged_batch_matches_per_sample_oracle():
    """VLMO-base at 480 px: two questions of 5 and 12 words in ONE batch (trimmed layout 14 + 901 = 915 tokens, key hole
    for the shorter question, 29 key tiles), the longer one with one substitutable word: joint attack through
    ``attack_mixed`` (image blocks + a text-gradient probe + substitution), each sample against its own batch-1 oracle
    loop at the full 941-token reference layout (vlmo_module.py:1387-1446, 1943-2055)."""
    from oracle import attack_loop
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    budget = 6
    cfg = vlmo_base(image_size=480)                       # what entry/run.py builds for task_finetune_vqa_base_image480
    assert cfg.n_image_tokens == 901
    cpu_model = FrozenVlmo(cfg, seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([5, 12], 40, seed=11)
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[1, 3] = True
    adapters = VlmoAttackAdapters(gpu_model)
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att, threshold=0)
    assert [len(p) for p in proposals] == [0, 1]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters, "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    res = attack.attack_mixed(img.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == 2 * budget + 1
    for s in range(2):
        adv, new_ids, losses = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img[s:s + 1], ids[s:s + 1],
                                                      masks[s:s + 1], proposals[s] if proposals[s] else None, sim,
                                                      init_eta=eta[s:s + 1], budget=budget, sim_threshold=0.3)
        assert res.adv_text_ids[s].cpu().tolist() == new_ids[0].tolist(), s
        _compare(res.adv_images[s].cpu(), adv[0].detach(), budget + 1)


def test_vlmo_base_480px_image_pgd_losses_match_cpu_oracle():
    """The operator-level twin of the 384-px 40-step test: image-only PGD, 8 steps, one 9-word question, losses 1e-4."""
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    steps = 8
    cpu_model = FrozenVlmo(vlmo_base(image_size=480), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([9], 40, seed=12)
    attack = BatchedVQAttack(VlmoAttackAdapters(gpu_model), "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=steps, sanity_checks=True))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV))
    ad = VlmoRefAdapters(cpu_model, ids, masks)
    with torch.enable_grad():
        adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img, EPS, EPS_ITER, steps, np.inf, clip_min=-1,
                                                        clip_max=1, y=ad.gen_ori_feats(img), ori_x=img, time=0, ls=1,
                                                        flavor="vlmo", init_eta=eta)
    _compare(res.adv_images[0].cpu(), adv[0].detach(), steps)
    np.testing.assert_allclose(res.loss_lists[0], losses, rtol=1e-4)


def test_vlmo_base_480px_full_40_step_attack_matches_cpu_oracle():
    """The reference's call at the reference's resolution with the reference's budget: 40 PGD steps at 480 px (941 tokens on
    the oracle side, 910 after the padding trim on the device).  Tolerances of a complete attack: >= 99 % of the pixels
    bit-identical, |dev| <= 4 eps_iter, mean |dev| <= 1e-4, losses 1e-4 relative."""
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import VlmoRefAdapters
    from tests.test_fullsize_parity import _compare as compare_full
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    steps = 40
    cpu_model = FrozenVlmo(vlmo_base(image_size=480), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([7], 40, seed=14)
    attack = BatchedVQAttack(VlmoAttackAdapters(gpu_model), "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=steps, sanity_checks=True))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV))
    ad = VlmoRefAdapters(cpu_model, ids, masks)
    with torch.enable_grad():
        adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img, EPS, EPS_ITER, steps, np.inf, clip_min=-1,
                                                        clip_max=1, y=ad.gen_ori_feats(img), ori_x=img, time=0, ls=1,
                                                        flavor="vlmo", init_eta=eta)
    same = compare_full(res.adv_images[0].cpu(), adv[0].detach(), steps, full_attack=True)
    print("VLMO-base 480 px, 40 steps: {:.3%} of the pixels bit-identical to the CPU oracle".format(same))
    assert len(res.loss_lists[0]) == steps
    np.testing.assert_allclose(res.loss_lists[0], losses, rtol=1e-4)


def test_albef_base_480px_attack_matches_cpu_oracle():
    """ALBEF-base at its configured ``image_res: 480``: ViT over 901 tokens, six fusion layers whose text queries attend
    901 image keys; 8 steps, batch 2 with a padded second question, against the per-sample oracle."""
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import AlbefRefAdapters
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base
    steps = 8
    cfg = albef_base(480, mlm_probability=0.0)
    assert cfg.n_image_tokens == 901
    cpu_model = FrozenAlbef(cfg, seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([10, 6], 12, seed=13)      # padding='longest': 12 tokens, the second question padded
    attack = BatchedVQAttack(AlbefAttackAdapters(gpu_model), "albef", gpu_model.embedding_tables(),
                             AttackConfig(budget=steps, sanity_checks=True))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV))
    total = np.zeros(steps)
    for s in range(2):
        n = int(masks[s].sum())
        ad = AlbefRefAdapters(cpu_model, ids[s:s + 1, :n], masks[s:s + 1, :n])
        tgt = ad.gen_ori_feats(img[s:s + 1])
        with torch.enable_grad():
            adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img[s:s + 1], EPS, EPS_ITER, steps, np.inf,
                                                            clip_min=-1, clip_max=1, y=[tgt[0], tgt[1], None, None, None],
                                                            ori_x=img[s:s + 1], time=0, ls=1, flavor="albef",
                                                            init_eta=eta[s:s + 1])
        _compare(res.adv_images[s].cpu(), adv[0].detach(), steps)
        total += np.asarray(losses)
    np.testing.assert_allclose(res.loss_lists[0], total, rtol=1e-4)     # a batch's loss is the sum of its samples'
