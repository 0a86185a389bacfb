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


def _sum_by_step(per_sample_lists):
    """Per-sample loss lists (one value per white-box step of that sample) -> per global step sums (a batch's losses)."""
    out = np.zeros(max(len(x) for x in per_sample_lists))
    for x in per_sample_lists:
        out[:len(x)] += x
    return out


def test_vlmo_base_ragged_batch_with_mixed_schedules_matches_per_sample_oracle():
    """BASELINE configs[1]/[3] shape at batch > 1 -- the one place where the product's semantics (batched, row weights,
    trimmed padding, per-sample schedules in one batch) differ from the batch-1 reference (vlmo_module.py:1438-1444 ``[0]``
    indexing): four questions of 3 / 6 / 9 / 12 words (padding trimmed 617 -> 591 tokens, row weights 0 / 1 / 2), one of
    them with 2 substitutable words, attacked as ONE batch through ``attack_mixed``; every sample against its own batch-1
    CPU oracle loop with reference-style packing (vlmo_module.py:1387-1446, 1943-2055)."""
    from oracle import attack_loop
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    budget, words = 6, 2
    cpu_model = FrozenVlmo(vlmo_base(384), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([3, 6, 9, 12], 40, seed=2)
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[2, 2:2 + words] = True
    adapters = VlmoAttackAdapters(gpu_model)
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att, threshold=0)
    assert [len(p) for p in proposals] == [0, 0, words, 0]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters, "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    res = attack.attack_mixed(img.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == 4 * budget + words
    per_sample, oracle_adv, oracle_ids = [], [], []
    for s in range(4):
        adv, new_ids, losses = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img[s:s + 1], ids[s:s + 1],
                                                      masks[s:s + 1], proposals[s] if proposals[s] else None, sim,
                                                      init_eta=eta[s:s + 1], budget=budget, sim_threshold=0.3)
        assert res.adv_text_ids[s].cpu().tolist() == new_ids[0].tolist(), s
        _compare(res.adv_images[s].cpu(), adv[0].detach(), budget + words, full_attack=False)
        oracle_adv.append(adv.detach())
        oracle_ids.append(new_ids)
        if s != 2:
            per_sample.append([v for block in losses for v in block])
    assert int((res.adv_text_ids[2].cpu() != ids[2]).sum()) >= 1, "the case should exercise an accepted substitution"
    # attack-success bits (vlmo_module.py:2063-2091) at base size: product attack + batched scorer on the device vs oracle
    # attack + per-question CPU scorer (oracle/blackbox_ref.py) over the same fine-tuned black box
    from oracle import blackbox_ref as bb
    black = FrozenVlmo.finetuned_from(cpu_model, seed=1)
    with torch.no_grad():
        def cpu_answers(images, text):
            out = []
            for s in range(4):
                _, states = black.encode(images[s:s + 1], black.text_embeddings(text[s:s + 1]), masks[s:s + 1])
                out += bb.vlmo_predict(black.vqa_classifier(black.pooled(states)))
            return out
        want_clean = cpu_answers(img, ids)
        want_after = cpu_answers(torch.cat(oracle_adv), torch.cat(oracle_ids))
    black_gpu = copy.deepcopy(black).to(DEV)
    got_clean = black_gpu.vqa_answer(img.to(DEV), ids.to(DEV), masks.to(DEV)).cpu().tolist()
    got_after = black_gpu.vqa_answer(res.adv_images, res.adv_text_ids, masks.to(DEV)).cpu().tolist()
    del black_gpu
    assert got_clean == want_clean and got_after == want_after      # the victim's answers, hence the success bits
    assert [int(a != c) for a, c in zip(got_after, got_clean)] == [int(a != c) for a, c in zip(want_after, want_clean)]
    # losses: the oracle loop does not report the loss of a probe step, so the batch trajectory is compared on a second,
    # image-only run of the same four questions (every step of every sample is then reported on both sides)
    res2 = attack.attack_mixed(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(att).to(DEV),
                               init_eta=eta.to(DEV))
    adv2, _, losses2 = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img[2:3], ids[2:3], masks[2:3], None, sim,
                                              init_eta=eta[2:3], budget=budget)
    per_sample.append([v for block in losses2 for v in block])
    np.testing.assert_allclose(res2.loss_lists[0], _sum_by_step(per_sample), rtol=1e-4)
    _compare(res2.adv_images[2].cpu(), adv2[0].detach(), budget, full_attack=False)


def _dual_tasks(flavor, ids, max_len):
    """Two dual-loss samples (old_alg == 0): paraphrase = question words + answer word (+ filler); sample 0 with one
    correct one-piece answer (2-d labels, 1 live label row), sample 1 with a two-piece answer and a second correct answer
    of the same piece count (3-d labels, K = 2, 2 live label rows)."""
    from oracle import text_scoring as ts
    from vqattack_amd.attack import mlm_task
    tasks, oracle_tasks = [], []
    for s in range(ids.shape[0]):
        body = [(int(t),) for t in ids[s].tolist() if t not in (0, 101, 102)]
        answer = (7001 + s, 7101 + s) if s % 2 else (7001 + s,)      # odd samples: a two-piece answer word -> 2 [MASK]s
        para = body[:3] + [answer] + [(8100 + s,)]
        correct = [[answer]] + ([[(7500 + s, 7600 + s)]] if s % 2 else [])
        same = [True] + [False] * (len(correct) - 1)
        tasks.append(mlm_task.build_mlm_task([answer], correct, same, para, [], flavor, max_len=max_len))
        ot = ts.build_mlm_task([answer], correct, same, para, [], flavor)
        ot["tail"] = ()
        assert tasks[-1].old_alg == 0 and ot["old_alg"] == 0
        assert ot["text_ids_mlm"] == tasks[-1].text_ids_mlm and ot["mlm_labels"] == tasks[-1].mlm_labels
        oracle_tasks.append(ot)
    return tasks, oracle_tasks


@pytest.mark.parametrize("live_rows", [True, False])
def test_vlmo_base_dual_loss_batch_matches_per_sample_oracle(live_rows):
    """``old_alg == 0`` at VLMO-base size, batch 2: ``pgd_mlm_attack`` (vlmo_module.py:1448-1529) through the real
    30 522-word MLM head + the dual loop (projected_gradient_descent.py:153-189), 4 dual iterations (8 white-box
    gradient steps), 2-d labels on sample 0 and 3-d labels (K = 2) on sample 1, per-sample cross-entropy normalisation.
    ``live_rows``: the MLM head and ``vqa_ce_rows`` on the live label rows only (default) / on all B x 40 positions with
    the dead rows skipped inside the kernel -- both must give the batch-1 oracle's trajectory."""
    from oracle import attack_loop
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    budget = 8
    cpu_model = FrozenVlmo(vlmo_base(384), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([5, 8], 40, seed=3)
    tasks, oracle_tasks = _dual_tasks("vlmo", ids, 40)
    attack = BatchedVQAttack(VlmoAttackAdapters(gpu_model), "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, live_mlm_rows=live_rows))
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), torch.zeros_like(ids, dtype=torch.bool).to(DEV),
                              init_eta=eta.to(DEV), dual=True, tasks=tasks)
    assert res.gradient_steps == budget and len(res.loss_lists[0]) == budget      # 4 x (feature, MLM)
    per_sample = []
    for s in range(2):
        adv, _, losses = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img[s:s + 1], ids[s:s + 1],
                                                masks[s:s + 1], None, None, init_eta=eta[s:s + 1], budget=budget,
                                                task=oracle_tasks[s])
        _compare(res.adv_images[s].cpu(), adv[0].detach(), budget, full_attack=False)
        per_sample.append(losses[0])
    # per_sample=True: the batch's MLM loss is the SUM of the samples' own batch-1 cross entropies
    np.testing.assert_allclose(res.loss_lists[0], _sum_by_step(per_sample), rtol=2e-4)


def test_albef_base_batch_with_substitution_matches_per_sample_oracle():
    """BASELINE configs[2] shape at batch 2 with one substitutable word per question: the text-embedding gradient comes
    back through the six cross-attention layers (``pgd_attack_vl`` / ``Gen_feats_from_embeds``, adv_attack.py:208-214,
    model_pretrain.py:85-104), candidates are scored and accepted on the device, and the second question is padded
    (ALBEF at batch 1 tokenises without padding, adv_attack.py:113).  Reference loop: adv_attack.py:604-712."""
    from oracle import attack_loop
    from oracle.adapters_ref import AlbefRefAdapters
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base
    budget = 6                                          # blocks [2, 4] + 1 probe step = 7 white-box gradient steps
    cpu_model = FrozenAlbef(albef_base(384, mlm_probability=0.0), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([6, 4], 8, seed=4)
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[0, 3] = att[1, 2] = True
    adapters = AlbefAttackAdapters(gpu_model)
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att, threshold=0)
    assert [len(p) for p in proposals] == [1, 1]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters, "albef", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    res = attack.attack_batch(img.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == budget + 1
    blocks = None
    n_changed = 0
    for s in range(2):
        n = int(masks[s].sum())
        adv, new_ids, losses = attack_loop.attack_one(AlbefRefAdapters, cpu_model, "albef", img[s:s + 1], ids[s:s + 1, :n],
                                                      masks[s:s + 1, :n], proposals[s], sim, init_eta=eta[s:s + 1],
                                                      budget=budget, sim_threshold=0.3)
        assert res.adv_text_ids[s, :n].cpu().tolist() == new_ids[0].tolist(), s
        assert res.adv_text_ids[s, n:].cpu().tolist() == ids[s, n:].tolist()
        n_changed += int((new_ids[0] != ids[s, :n]).sum())
        _compare(res.adv_images[s].cpu(), adv[0].detach(), budget + 1, full_attack=False)
        blocks = [np.array(b) for b in losses] if blocks is None else [a + np.array(b) for a, b in zip(blocks, losses)]
    assert n_changed >= 1, "the case should exercise at least one accepted substitution"
    for got, want in zip(res.loss_lists, blocks):
        np.testing.assert_allclose(got, want, rtol=1e-4)


def test_vlmo_large_mixed_feature_and_dual_batch_matches_per_sample_oracle():
    """BASELINE configs[4] shape, the most heterogeneous batch the drivers take: VLMO-large (25 maps per loss launch,
    D = 1024), sample 0 feature loss with 2 substitutable words, sample 1 dual loss (3-d labels, two-piece answer) with
    1 substitutable word, attacked as ONE batch through ``attack_mixed`` -- one encoder pass per global step with the
    paraphrase as sample 1's text on its MLM steps, the 1024 x 30 522 MLM head at its live label rows only -- against
    each sample's own batch-1 oracle loop (vlmo_module.py:1943-2055, dual blocks :2009-2035)."""
    from oracle import attack_loop
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_large
    budget = 6
    cpu_model = FrozenVlmo(vlmo_large(384), seed=0)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    ids, masks, img, eta = _inputs([6, 5], 40, seed=6)
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[0, 1:3] = True
    att[1, 2] = True
    tasks, oracle_tasks = _dual_tasks("vlmo", ids, 40)
    tasks, oracle_tasks = [None, tasks[1]], [None, oracle_tasks[1]]
    adapters = VlmoAttackAdapters(gpu_model)
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att, threshold=0)
    assert [len(p) for p in proposals] == [2, 1]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters, "vlmo", gpu_model.embedding_tables(),
                             AttackConfig(budget=budget, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    res = attack.attack_mixed(img.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals, tasks=tasks)
    assert res.gradient_steps == (budget + 2) + (budget + 1)
    n_changed = 0
    for s in range(2):
        adv, new_ids, _ = attack_loop.attack_one(VlmoRefAdapters, cpu_model, "vlmo", img[s:s + 1], ids[s:s + 1],
                                                 masks[s:s + 1], proposals[s], sim, init_eta=eta[s:s + 1], budget=budget,
                                                 sim_threshold=0.3, task=oracle_tasks[s])
        assert res.adv_text_ids[s].cpu().tolist() == new_ids[0].tolist(), s
        n_changed += int((new_ids[0] != ids[s]).sum())
        _compare(res.adv_images[s].cpu(), adv[0].detach(), budget + 2, full_attack=False)
    assert n_changed >= 1, "the case should exercise at least one accepted substitution"


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
