"""Attack-success BITS of the product pipeline vs the oracle pipeline (SURVEY.md section 8 row a12 + the judge's row g).

The reference's end product is one bit per sample: ``acc_list.append(1 if the victim's answer changed else 0)``
(``ALBEF_attack/adv_attack.py:717-733``; ``vlmo/modules/vlmo_module.py:2063-2091``), and ``north_star`` asks for an attack
success rate within +-0.5 % of the reference's.  Here, for the same seeded samples and the same frozen networks:

  product : batched joint attack on the MI355X (``attack_mixed``: HIP operators, per-sample schedules and loss modes in
            one batch, acceptance on the device) + the batched black-box scorer (``vqa_answer``: answer classifier /
            ``rank_answer``) on the device;
  oracle  : the per-sample CPU loop (``oracle/attack_loop.attack_one``, pinned by the reference's own loop code) + the
            per-question CPU scorers (``oracle/blackbox_ref``, ``rank_answer`` pinned by the reference's method).

Required: the two success-bit vectors are EQUAL (a disagreement budget of 0.5 % of 32 samples is zero samples), the
clean answers and the answers to the adversarial pairs are equal, and the set is informative (both outcomes occur).
Perturbations that differ in < 1 % of the pixels by 2 * eps_iter (sign flips of ~0 gradients, see test_fullsize_parity)
must not move a decision; the margins of the oracle's decisions are printed so that a failure can be told from a
borderline sample.
"""
import contextlib

import numpy as np
import pytest
import torch

N_SAMPLES = 32
BUDGET = 10


def make_samples(flavor, cfg, n=N_SAMPLES, seed=7, words=(2, 6), max_att=4, text_len=None):
    """Seeded questions (``words[0]`` .. ``words[1]`` - 1 words, padded; default 2..5), 0 .. ``max_att`` - 1 substitutable
    words each (default 0..3), every 4th sample dual-loss (old_alg == 0) with a paraphrase; returns (ids, masks,
    attackable, product tasks, oracle tasks, images, eta)."""
    from oracle import text_scoring as ts
    from vqattack_amd.attack import mlm_task
    text_len = text_len or (cfg.max_text_len if flavor == "vlmo" else 8)
    r = np.random.RandomState(seed)
    ids = torch.zeros(n, text_len, dtype=torch.long)
    att = torch.zeros(n, text_len, dtype=torch.bool)
    tasks, oracle_tasks = [], []
    for s in range(n):
        k = int(r.randint(words[0], words[1]))
        ids[s, 0], ids[s, 1 + k] = 101, 102
        ids[s, 1:1 + k] = torch.from_numpy(r.randint(1000, 30522, k))
        n_att = int(r.randint(0, max_att))
        for p in sorted(r.choice(np.arange(1, 1 + k), size=min(n_att, k), replace=False).tolist()):
            att[s, p] = True
        if s % 4 == 0:
            body = [(int(t),) for t in ids[s, 1:1 + k].tolist()]
            answer = (int(r.randint(1000, 30522)),)
            para = body[:2] + [answer]
            correct = [[answer]] + ([[(int(r.randint(1000, 30522)),)]] if s % 8 == 0 else [])
            same = [True] + [False] * (len(correct) - 1)
            max_len = text_len if flavor == "vlmo" else None
            tasks.append(mlm_task.build_mlm_task([answer], correct, same, para, [], flavor, max_len=max_len))
            ot = ts.build_mlm_task([answer], correct, same, para, [], flavor)
            if flavor == "vlmo":        # the oracle encodes at the reference's literal 40: re-encode at this model's length
                ot["text_ids_mlm"], ot["text_mask_mlm"] = ts.encode_words(ot["list_words"], text_len, text_len)
                lab = ot["mlm_labels"]
                ot["mlm_labels"] = [row[:text_len] for row in lab] if isinstance(lab[0], list) else lab[:text_len]
            ot["tail"] = ()
            assert ot["text_ids_mlm"] == tasks[-1].text_ids_mlm and ot["mlm_labels"] == tasks[-1].mlm_labels
            oracle_tasks.append(ot)
        else:
            tasks.append(None)
            oracle_tasks.append(None)
    g = torch.Generator().manual_seed(seed)
    images = torch.empty(n, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    return ids, (ids != 0).long(), att, tasks, oracle_tasks, images, eta


def build(flavor, size="tiny", **cfg_kw):
    """(white box, black box = its fine-tuned copy with the VQA head, adapters class, reference adapters class, cfg).
    ``cfg_kw`` overrides fields of the model config (e.g. ``n_answers``: the size of the victim's answer list)."""
    from oracle.adapters_ref import AlbefRefAdapters, VlmoRefAdapters
    if flavor == "vlmo":
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base, vlmo_tiny
        cfg = vlmo_tiny(**cfg_kw) if size == "tiny" else vlmo_base(384, **cfg_kw)
        white = FrozenVlmo(cfg, seed=3)
        return white, FrozenVlmo.finetuned_from(white, seed=4), VlmoAttackAdapters, VlmoRefAdapters, cfg
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base, albef_tiny
    cfg = albef_tiny(mlm_probability=0.0, **cfg_kw) if size == "tiny" else albef_base(384, mlm_probability=0.0, **cfg_kw)
    white = FrozenAlbef(cfg, seed=3)
    return white, FrozenAlbef.finetuned_from(white, seed=4), AlbefAttackAdapters, AlbefRefAdapters, cfg


@contextlib.contextmanager
def oracle_text_len(flavor, text_len):
    """The oracle loop re-encodes a dual-loss paraphrase at the reference's literal 40 tokens (vlmo_module.py:1708);
    the test-sized VLMo has a shorter text: re-encode at its length."""
    from oracle import text_scoring as ts
    orig = ts.encode_words
    if flavor == "vlmo" and text_len != 40:
        ts.encode_words = lambda words, _ml, _pad, tail=(): orig(words, text_len, text_len, tail)
    try:
        yield
    finally:
        ts.encode_words = orig


def oracle_answers(flavor, black, images, ids, masks, n_answers=None):
    """Per-question CPU scorer (oracle/blackbox_ref.py); returns (answer index, decision margin) per sample.
    ``n_answers`` (VLMo): the victim answers from the first ``n_answers`` classes of its answer vocabulary only."""
    from oracle import blackbox_ref as bb
    answers, margins = [], []
    with torch.no_grad():
        for b in range(images.shape[0]):
            n = int(masks[b].sum()) if flavor == "albef" else ids.shape[1]
            qi, qm = ids[b:b + 1, :n], masks[b:b + 1, :n]
            if flavor == "vlmo":
                _, states = black.encode(images[b:b + 1], black.text_embeddings(qi), qm)
                logits = black.vqa_classifier(black.pooled(states))[:, :n_answers]
                answers += bb.vlmo_predict(logits)
                top = logits[0].topk(2).values
                margins.append(float(top[0] - top[1]))
            else:
                cfg = black.cfg
                st, _ = black.visual_encoder(images[b:b + 1])
                q, _ = black.text_encoder(black.text_embeddings(qi), qm, st)
                ans = black.answer_ids
                ti, tp = bb.rank_answer(black._decode, q, qm, ans, (ans != cfg.pad_id).long(),
                                        min(cfg.k_test, cfg.n_answers), pad_id=cfg.pad_id)
                answers += bb.albef_predict(ti, tp)
                margins.append(float(tp[0, 0] - tp[0, 1]))
    return answers, margins


def oracle_pipeline(flavor, white, black, ref_cls, cfg, samples, proposals, sim, budget=BUDGET):
    """(clean answers, adversarial answers, success bits, margins of the adversarial decisions) of the CPU oracle."""
    from oracle import attack_loop
    ids, masks, att, _, oracle_tasks, images, eta = samples
    clean, _ = oracle_answers(flavor, black, images, ids, masks)
    adv_images, adv_ids = [], ids.clone()
    with oracle_text_len(flavor, ids.shape[1]):
        for s in range(ids.shape[0]):
            n = int(masks[s].sum()) if flavor == "albef" else ids.shape[1]
            adv, new_ids, _ = attack_loop.attack_one(ref_cls, white, flavor, images[s:s + 1], ids[s:s + 1, :n],
                                                     masks[s:s + 1, :n], proposals[s] if proposals[s] else None, sim,
                                                     init_eta=eta[s:s + 1], budget=budget, sim_threshold=0.3,
                                                     task=oracle_tasks[s])
            adv_images.append(adv.detach())
            adv_ids[s, :n] = new_ids[0]
    after, margins = oracle_answers(flavor, black, torch.cat(adv_images), adv_ids, masks)
    bits = [int(a != c) for a, c in zip(after, clean)]
    return clean, after, bits, margins, adv_ids


@pytest.mark.gpu
@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_success_bits_equal_oracle_pipeline(flavor):
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    dev = torch.device("cuda", 0)
    white, black, adapters_cls, ref_cls, cfg = build(flavor)
    samples = make_samples(flavor, cfg)
    ids, masks, att, tasks, _, images, eta = samples
    import copy
    white_gpu, black_gpu = copy.deepcopy(white).to(dev), copy.deepcopy(black).to(dev)
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters_cls(white_gpu), flavor, white_gpu.embedding_tables(),
                             AttackConfig(budget=BUDGET, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    # candidate proposals: computed once on the device, injected on both sides (host data)
    proposals = text_update.propose_candidates(attack.adapters.mlm_logits(ids.to(dev), masks.to(dev)), ids, att,
                                               threshold=0)
    # ---- product: two mixed batches of 16 (different schedules and loss modes inside each), scorer on the device
    got_clean = black_gpu.vqa_answer(images.to(dev), ids.to(dev), masks.to(dev)).cpu().tolist()
    got_after, got_ids = [], []
    for lo in range(0, N_SAMPLES, 16):
        sl = slice(lo, lo + 16)
        res = attack.attack_mixed(images[sl].to(dev), ids[sl].to(dev), masks[sl].to(dev), att[sl].to(dev),
                                  init_eta=eta[sl].to(dev), proposals=proposals[sl], tasks=tasks[sl])
        got_after += black_gpu.vqa_answer(res.adv_images, res.adv_text_ids, masks[sl].to(dev)).cpu().tolist()
        got_ids.append(res.adv_text_ids.cpu())
    got_bits = [int(a != c) for a, c in zip(got_after, got_clean)]
    # ---- oracle: per-sample CPU loop, per-question CPU scorer
    clean, after, bits, margins, adv_ids = oracle_pipeline(flavor, white, black, ref_cls, cfg, samples, proposals, sim)
    print("oracle success bits", bits, "ASR", np.mean(bits))
    print("smallest oracle decision margins", sorted(margins)[:4])
    assert got_clean == clean
    assert torch.equal(torch.cat(got_ids), adv_ids), "substituted token ids differ"
    assert got_bits == bits, [(s, got_after[s], after[s], margins[s]) for s in range(N_SAMPLES)
                              if got_bits[s] != bits[s]]
    # stronger than the bits: the victim's ANSWER to every adversarial pair is the same on both sides
    assert got_after == after, [(s, got_after[s], after[s], margins[s]) for s in range(N_SAMPLES)
                                if got_after[s] != after[s]]
    assert 0 < sum(bits) < N_SAMPLES, "the sample set should contain successes and failures"


def test_oracle_pipeline_is_informative():
    """CPU-only guard for the sample set above: the oracle pipeline alone must produce both outcomes for both flavors
    (a set on which every attack fails, or every attack succeeds, would make bit equality vacuous)."""
    from vqattack_amd.attack import text_update
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    for flavor in ("vlmo", "albef"):
        white, black, _, ref_cls, cfg = build(flavor)
        samples = make_samples(flavor, cfg, n=12)
        proposals = [[] for _ in range(12)]              # image-only here: candidate proposals need the device
        ids, masks = samples[0], samples[1]
        _, _, bits, _, _ = oracle_pipeline(flavor, white, black, ref_cls, cfg, samples, proposals, sim, budget=BUDGET)
        assert 0 < sum(bits) < 12, (flavor, bits)
