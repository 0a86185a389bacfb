"""Batched joint attack on the MI355X vs the per-sample (batch 1, reference-style packing) CPU oracle.

Same frozen white box (seeded weights) on both sides; the product attacks a batch of 3 questions with different
padding in one go (LayerFeatures + row weights, no packing), the oracle attacks them one at a time the way the
reference does.  Tolerances: >= 99 % of the pixels bit-identical per sample (sign flips of ~0 gradients aside, every
step moves a pixel by exactly +-eps_iter), substituted token ids identical, summed per-sample losses within 2e-4
relative of the batch loss.
"""
import numpy as np
import pytest
import torch

from oracle import attack_loop
from oracle.adapters_ref import AlbefRefAdapters, VlmoRefAdapters
from vqattack_amd.attack import text_update

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

IDS = torch.tensor([[101, 2054, 3609, 2003, 102, 0, 0, 0],
                    [101, 2129, 2116, 6077, 2024, 102, 0, 0],
                    [101, 2003, 2009, 102, 0, 0, 0, 0]])
ATTACKABLE = torch.zeros_like(IDS, dtype=torch.bool)
ATTACKABLE[0, [1, 2]] = True
ATTACKABLE[1, [2, 4]] = True
ATTACKABLE[2, [1, 2]] = True


def _build(flavor):
    if flavor == "vlmo":
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
        cfg = vlmo_tiny()
        return (FrozenVlmo(cfg, seed=3), FrozenVlmo(cfg, seed=3).to(DEV), VlmoAttackAdapters, VlmoRefAdapters, cfg)
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_tiny
    cfg = albef_tiny(mlm_probability=0.0)      # the per-forward random token masking is exercised in test_albef_masking
    return (FrozenAlbef(cfg, seed=3), FrozenAlbef(cfg, seed=3).to(DEV), AlbefAttackAdapters, AlbefRefAdapters, cfg)


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
@pytest.mark.parametrize("with_words", [True, False])
def test_batched_attack_matches_per_sample_oracle(flavor, with_words):
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(11)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    masks = (IDS != 0).long()
    attackable = ATTACKABLE if with_words else torch.zeros_like(ATTACKABLE)
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=10, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    # candidate proposals: computed once on the device, injected on both sides (host data)
    proposals = None
    if with_words:
        logits = attack.adapters.mlm_logits(IDS.to(DEV), masks.to(DEV))
        proposals = text_update.propose_candidates(logits, IDS, attackable, threshold=0)
        assert all(len(p) == 2 for p in proposals)
    res = attack.attack_batch(images.to(DEV), IDS.to(DEV), masks.to(DEV), attackable.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == (10 + 2 if with_words else 10)
    want_losses = None
    for s in range(3):
        # ALBEF at batch 1 tokenises with padding='longest', i.e. WITHOUT pad tokens (adv_attack.py:113); VLMo pads every
        # question to max_text_len and drops the padded rows through the mask (vlmo_module.py:1440-1442)
        n = int(masks[s].sum()) if flavor == "albef" else IDS.shape[1]
        adv, ids, losses = attack_loop.attack_one(ref_cls, cpu_model, flavor, images[s:s + 1], IDS[s:s + 1, :n],
                                                  masks[s:s + 1, :n], proposals[s] if with_words else None, sim,
                                                  init_eta=eta[s:s + 1], budget=10, sim_threshold=0.3)
        same = (res.adv_images[s].cpu() == adv[0]).float().mean().item()
        assert same >= 0.99, (flavor, s, same)
        assert float((res.adv_images[s].cpu() - images[s]).abs().max()) <= np.float32(0.125) + 1e-7
        assert res.adv_text_ids[s, :n].cpu().tolist() == ids[0].tolist(), (flavor, s)
        assert res.adv_text_ids[s, n:].cpu().tolist() == IDS[s, n:].tolist()
        flat = np.array([v for block in losses for v in block])
        want_losses = flat if want_losses is None else want_losses + flat
    got_losses = np.array([v for block in res.loss_lists for v in block])
    assert np.allclose(got_losses, want_losses, rtol=2e-4, atol=1e-5), (got_losses, want_losses)
    if with_words:
        changed = (res.adv_text_ids.cpu() != IDS).sum().item()
        assert changed >= 1, "the test is meant to exercise at least one accepted substitution"


def test_albef_masking_is_reproducible_and_respects_specials():
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_tiny
    m = FrozenAlbef(albef_tiny(mlm_probability=0.5), seed=0)
    ids = IDS.clone()
    m.seed_masking(7)
    a = m.mask_tokens(ids)
    m.seed_masking(7)
    b = m.mask_tokens(ids)
    assert torch.equal(a, b) and not torch.equal(a, ids)
    assert torch.equal(a[:, 0], ids[:, 0])                       # [CLS] never masked
    assert torch.equal(a[ids == 0], ids[ids == 0])               # padding never masked
    assert torch.equal(ids, IDS)                                 # input untouched


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_batched_dual_loss_attack_matches_per_sample_oracle(flavor):
    """Dual-loss blocks (feature step + MLM step per iteration, `old_alg == 0` in the reference).  The MLM cross entropy
    is a mean over the valid tokens of the whole batch here and of one sample in the reference: a positive per-sample
    scale that `sign()` cannot see, so the perturbations must still agree; loss VALUES of the MLM steps are not compared."""
    import numpy as np
    from oracle import cleverhans_cpu as o
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(21)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    masks = (IDS != 0).long()
    labels = torch.full_like(IDS, -100)
    labels[0, 2], labels[1, 3], labels[2, 1] = 2003, 2024, 2054            # one masked word per question
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=8, sanity_checks=True))
    res = attack.attack_batch(images.to(DEV), IDS.to(DEV), masks.to(DEV), torch.zeros_like(ATTACKABLE).to(DEV),
                              dual=True, mlm_labels=labels.to(DEV), init_eta=eta.to(DEV))
    assert len(res.loss_lists[0]) == 8                                   # 4 iterations x (feature, MLM)
    for s in range(3):
        n = int(masks[s].sum()) if flavor == "albef" else IDS.shape[1]
        ad = ref_cls(cpu_model, IDS[s:s + 1, :n], masks[s:s + 1, :n])
        targets = ad.gen_ori_feats(images[s:s + 1])
        lab = labels[s:s + 1, :n]
        y = [lab, targets[0], targets[1]] if flavor == "albef" else [lab, targets[1], targets[2]]
        with torch.enable_grad():
            adv, _ = o.projected_gradient_descent([ad.pgd_attack, ad.pgd_mlm_attack], images[s:s + 1], 0.125, 0.01, 4,
                                                  np.inf, clip_min=-1, clip_max=1, y=y, ori_x=images[s:s + 1], time=0,
                                                  ls=0, flavor=flavor, init_eta=eta[s:s + 1])
        same = (res.adv_images[s].cpu() == adv[0]).float().mean().item()
        assert same >= 0.99, (flavor, s, same)


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_mixed_schedule_batch_matches_per_sample_oracle(flavor):
    """One batch, three different schedules (0, 2 and 3 substitutable words): prefix scheduling must give every sample the
    result of its own batch-1 attack."""
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(31)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    masks = (IDS != 0).long()
    att = torch.zeros_like(ATTACKABLE)
    att[1, [1, 2, 4]] = True                      # sample 1: three words, sample 2: two, sample 0: none
    att[2, [1, 2]] = True
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=10, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    logits = attack.adapters.mlm_logits(IDS.to(DEV), masks.to(DEV))
    proposals = text_update.propose_candidates(logits, IDS, att, threshold=0)
    res = attack.attack_mixed(images.to(DEV), IDS.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    assert res.gradient_steps == 10 + 13 + 12
    for s in range(3):
        n = int(masks[s].sum()) if flavor == "albef" else IDS.shape[1]
        adv, ids, _ = attack_loop.attack_one(ref_cls, cpu_model, flavor, images[s:s + 1], IDS[s:s + 1, :n],
                                             masks[s:s + 1, :n], proposals[s] if proposals[s] else None, sim,
                                             init_eta=eta[s:s + 1], budget=10, sim_threshold=0.3)
        same = (res.adv_images[s].cpu() == adv[0]).float().mean().item()
        assert same >= 0.99, (flavor, s, same)
        assert res.adv_text_ids[s, :n].cpu().tolist() == ids[0].tolist(), (flavor, s)
    assert (res.adv_text_ids.cpu() != IDS).sum().item() >= 1


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_batched_dual_attack_with_ragged_mlm_tasks_matches_per_sample_oracle(flavor):
    """Dual-loss (old_alg == 0) batch built from ``MlmTask``s: three samples whose paraphrases have different lengths
    and whose label sets differ in number (K = 1, 2, 3 -> padded with all-ignored sets), with the paraphrase following
    the question's substitutions (``update_mlm_text``).  Per-sample cross-entropy normalisation makes every sample's
    trajectory equal to the batch-1 oracle's, whose loop is pinned against the reference's own (test_text_golden_loops)."""
    from oracle import text_scoring as ts
    from vqattack_amd.attack import mlm_task
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(21)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    masks = (IDS != 0).long()
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    # per sample: paraphrase = some question words + extra words + the answer word; 1..3 correct answers of one piece
    body = [[(int(t),) for t in IDS[s].tolist() if t not in (0, 101, 102)] for s in range(3)]
    answers = [(7001,), (7002,), (7003,)]
    paras = [body[0] + [answers[0]], body[1][:2] + [answers[1]] + [(8123,)], [answers[2]] + body[2] + [(8456,), (8457,)]]
    alts = [[], [(7102,)], [(7103,), (7203,)]]
    max_len = cfg.max_text_len if flavor == "vlmo" else None
    tasks, oracle_tasks = [], []
    for s in range(3):
        correct = [[answers[s]]] + [[a] for a in alts[s]]
        same = [True] + [False] * len(alts[s])
        tasks.append(mlm_task.build_mlm_task([answers[s]], correct, same, paras[s], [], flavor, max_len=max_len))
        assert tasks[-1].old_alg == 0
        ot = ts.build_mlm_task([answers[s]], correct, same, paras[s], [], flavor)
        if flavor == "vlmo":            # the oracle encodes at the reference's literal 40: re-encode at this model's length
            ot["text_ids_mlm"], ot["text_mask_mlm"] = ts.encode_words(ot["list_words"], max_len, max_len)
            ot["mlm_labels"] = [row[:max_len] for row in ot["mlm_labels"]] if isinstance(ot["mlm_labels"][0], list) \
                else ot["mlm_labels"][:max_len]
        assert ot["text_ids_mlm"] == tasks[-1].text_ids_mlm and ot["mlm_labels"] == tasks[-1].mlm_labels
        oracle_tasks.append(ot)
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=12, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    logits = attack.adapters.mlm_logits(IDS.to(DEV), masks.to(DEV))
    proposals = text_update.propose_candidates(logits, IDS, ATTACKABLE, threshold=0)
    # make the paraphrase share a question word with a candidate substitution, so that update_mlm_text has work to do
    res = attack.attack_batch(images.to(DEV), IDS.to(DEV), masks.to(DEV), ATTACKABLE.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals, dual=True, tasks=tasks)
    n_changed = 0
    for s in range(3):
        n = int(masks[s].sum()) if flavor == "albef" else IDS.shape[1]
        ot = dict(oracle_tasks[s])
        ot["tail"] = ()

        def factory(model, ids, msk, ids_mlm=None, msk_mlm=None):
            return ref_cls(model, ids, msk, ids_mlm, msk_mlm)
        if flavor == "vlmo":            # the oracle loop re-encodes the paraphrase at 40: keep this model's text length
            import oracle.text_scoring as _ts
            orig = _ts.encode_words
            _ts.encode_words = lambda words, _ml, _pad, tail=(): orig(words, max_len, max_len, tail)
        try:
            adv, ids, _ = attack_loop.attack_one(factory, cpu_model, flavor, images[s:s + 1], IDS[s:s + 1, :n],
                                                 masks[s:s + 1, :n], proposals[s], sim, init_eta=eta[s:s + 1],
                                                 budget=12, sim_threshold=0.3, task=ot)
        finally:
            if flavor == "vlmo":
                _ts.encode_words = orig
        assert res.adv_text_ids[s, :n].cpu().tolist() == ids[0].tolist(), (flavor, s)
        n_changed += int((ids[0] != IDS[s, :n]).sum())
        same_px = (res.adv_images[s].cpu() == adv[0]).float().mean().item()
        assert same_px >= 0.99, (flavor, s, same_px)
        mlm_len = len(tasks[s].text_ids_mlm)
        got_mlm = res.adv_text_ids_mlm[s, :mlm_len].cpu().tolist()
        want_words = list(oracle_tasks[s]["list_words"])
        first = [(int(a), int(b)) for a, b in zip(IDS[s, :n].tolist(), ids[0].tolist()) if a != b]
        # (the oracle's own paraphrase is internal to attack_one; rebuild it from the substitutions it made)
        if len(first) == len(set(a for a, _ in first)):
            ts.update_mlm_text(first, want_words)
            want_ids, _ = ts.encode_words(want_words, max_len or 25, max_len, ())
            assert got_mlm == want_ids[:mlm_len]
    assert n_changed >= 1, "the case should exercise at least one accepted substitution"


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_mixed_batch_of_feature_and_dual_samples_matches_per_sample_oracle(flavor):
    """``attack_mixed`` with per-sample loss modes: sample 0 feature loss with 2 words, sample 1 dual loss (3-d labels,
    K = 2) with 2 words, sample 2 dual loss (K = 3) without substitutable words -- three different step sequences in one
    white-box batch -- against the batch-1 CPU oracle loop of each sample."""
    from oracle import text_scoring as ts
    from vqattack_amd.attack import mlm_task
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(41)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    masks = (IDS != 0).long()
    att = ATTACKABLE.clone()
    att[2] = False
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    body = [[(int(t),) for t in IDS[s].tolist() if t not in (0, 101, 102)] for s in range(3)]
    answers = [None, (7002,), (7003,)]
    paras = [None, body[1][:2] + [answers[1]] + [(8123,)], [answers[2]] + body[2] + [(8456,), (8457,)]]
    alts = [None, [(7102,)], [(7103,), (7203,)]]
    max_len = cfg.max_text_len if flavor == "vlmo" else None
    tasks, oracle_tasks = [None], [None]
    for s in (1, 2):
        correct = [[answers[s]]] + [[a] for a in alts[s]]
        same = [True] + [False] * len(alts[s])
        tasks.append(mlm_task.build_mlm_task([answers[s]], correct, same, paras[s], [], flavor, max_len=max_len))
        assert tasks[-1].old_alg == 0
        ot = ts.build_mlm_task([answers[s]], correct, same, paras[s], [], flavor)
        if flavor == "vlmo":
            ot["text_ids_mlm"], ot["text_mask_mlm"] = ts.encode_words(ot["list_words"], max_len, max_len)
            ot["mlm_labels"] = [row[:max_len] for row in ot["mlm_labels"]]
        ot["tail"] = ()
        oracle_tasks.append(ot)
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=12, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    logits = attack.adapters.mlm_logits(IDS.to(DEV), masks.to(DEV))
    proposals = text_update.propose_candidates(logits, IDS, att, threshold=0)
    res = attack.attack_mixed(images.to(DEV), IDS.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals, tasks=tasks)
    assert res.gradient_steps == 14 + 14 + 12
    assert res.adv_text_ids.shape == IDS.shape
    n_changed = 0
    for s in range(3):
        n = int(masks[s].sum()) if flavor == "albef" else IDS.shape[1]
        if flavor == "vlmo":            # the oracle loop re-encodes the paraphrase at 40: keep this model's text length
            orig = ts.encode_words
            ts.encode_words = lambda words, _ml, _pad, tail=(): orig(words, max_len, max_len, tail)
        try:
            adv, ids, _ = attack_loop.attack_one(ref_cls, cpu_model, flavor, images[s:s + 1], IDS[s:s + 1, :n],
                                                 masks[s:s + 1, :n], proposals[s] if proposals[s] else None, sim,
                                                 init_eta=eta[s:s + 1], budget=12, sim_threshold=0.3,
                                                 task=oracle_tasks[s])
        finally:
            if flavor == "vlmo":
                ts.encode_words = orig
        assert res.adv_text_ids[s, :n].cpu().tolist() == ids[0].tolist(), (flavor, s)
        n_changed += int((ids[0] != IDS[s, :n]).sum())
        same_px = (res.adv_images[s].cpu() == adv[0]).float().mean().item()
        assert same_px >= 0.99, (flavor, s, same_px)
    assert n_changed >= 1


@pytest.mark.parametrize("with_words", [False, True])
def test_albef_random_token_masking_on_matches_oracle_at_batch_1(with_words):
    """ALBEF re-draws a random 15 % MLM mask inside EVERY white-box forward (model_pretrain.py:130-132, ``mask`` :309-332).
    With the mask generator seeded identically on both sides and batch 1 -- the reference's own batch size -- the product
    and the oracle issue the same sequence of draws (``Gen_ori_feats``, then one per PGD step, one drawn-and-ignored per
    probe step, :85-104), so the attack must agree as closely as with masking off.  For batch > 1 the product draws one
    (B, L) mask where the reference would draw B separate (1, L) masks: those runs are only statistically equivalent."""
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_tiny
    cfg = albef_tiny(mlm_probability=0.5)            # half of the tokens: every forward really changes the text
    cpu_model, gpu_model = FrozenAlbef(cfg, seed=3), FrozenAlbef(cfg, seed=3).to(DEV)
    g = torch.Generator().manual_seed(51)
    image = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(image).uniform_(-0.125, 0.125, generator=g)
    ids = IDS[1:2, :6]
    masks = (ids != 0).long()
    att = torch.zeros_like(ids, dtype=torch.bool)
    if with_words:
        att[0, [2, 4]] = True
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(AlbefAttackAdapters(gpu_model), "albef", gpu_model.embedding_tables(),
                             AttackConfig(budget=10, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    proposals = None
    if with_words:
        proposals = text_update.propose_candidates(attack.adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att,
                                                   threshold=0)
    gpu_model.seed_masking(7)
    res = attack.attack_batch(image.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    cpu_model.seed_masking(7)
    adv, new_ids, losses = attack_loop.attack_one(AlbefRefAdapters, cpu_model, "albef", image, ids, masks,
                                                  proposals[0] if with_words else None, sim, init_eta=eta, budget=10,
                                                  sim_threshold=0.3)
    assert res.adv_text_ids[0].cpu().tolist() == new_ids[0].tolist()
    same = (res.adv_images[0].cpu() == adv[0]).float().mean().item()
    assert same >= 0.99, same
    got = np.array([v for block in res.loss_lists for v in block])
    want = np.array([v for block in losses for v in block])
    assert np.allclose(got, want, rtol=2e-4, atol=1e-5), (got, want)
    # the masking really acted: the same attack with masking off takes another trajectory
    off = FrozenAlbef(albef_tiny(mlm_probability=0.0), seed=3).to(DEV)
    res_off = BatchedVQAttack(AlbefAttackAdapters(off), "albef", off.embedding_tables(),
                              AttackConfig(budget=10, sim_threshold=0.3), similarity_fn=sim).attack_batch(
        image.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV), proposals=proposals)
    assert (res_off.adv_images != res.adv_images).float().mean().item() > 0.01


def test_albef_random_token_masking_on_in_mixed_batches_matches_oracle_at_batch_1():
    """``attack_mixed`` feeds text EMBEDDINGS on every step, so the per-forward random token masking of the id-taking
    closures (``Gen_feats`` / ``get_mlm_logits``, model_pretrain.py:130-132,105-122) is applied by the driver: freshly masked
    ids are embedded for every sample that is not at a probe step (the probe closure ignores its draw, :85-104).  At batch 1
    the draws line up one to one with the reference-structured oracle loop: same seed, same attack."""
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_tiny
    cfg = albef_tiny(mlm_probability=0.5)
    cpu_model, gpu_model = FrozenAlbef(cfg, seed=3), FrozenAlbef(cfg, seed=3).to(DEV)
    g = torch.Generator().manual_seed(61)
    image = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(image).uniform_(-0.125, 0.125, generator=g)
    ids = IDS[1:2, :6]
    masks = (ids != 0).long()
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[0, [2, 4]] = True
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    attack = BatchedVQAttack(AlbefAttackAdapters(gpu_model), "albef", gpu_model.embedding_tables(),
                             AttackConfig(budget=10, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    proposals = text_update.propose_candidates(attack.adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att,
                                               threshold=0)
    gpu_model.seed_masking(9)
    res = attack.attack_mixed(image.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV),
                              proposals=proposals)
    cpu_model.seed_masking(9)
    adv, new_ids, _ = attack_loop.attack_one(AlbefRefAdapters, cpu_model, "albef", image, ids, masks, proposals[0], sim,
                                             init_eta=eta, budget=10, sim_threshold=0.3)
    assert res.adv_text_ids[0].cpu().tolist() == new_ids[0].tolist()
    same = (res.adv_images[0].cpu() == adv[0]).float().mean().item()
    assert same >= 0.99, same
    # and the masking acted: with the same seed but masking off the trajectory differs
    off = FrozenAlbef(albef_tiny(mlm_probability=0.0), seed=3).to(DEV)
    res_off = BatchedVQAttack(AlbefAttackAdapters(off), "albef", off.embedding_tables(),
                              AttackConfig(budget=10, sim_threshold=0.3), similarity_fn=sim).attack_mixed(
        image.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV), init_eta=eta.to(DEV), proposals=proposals)
    assert (res_off.adv_images != res.adv_images).float().mean().item() > 0.01


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_mixed_batch_of_only_dual_samples_equals_the_bucket_path_and_the_oracle(flavor):
    """Every sample dual-loss with the same schedule: on the MLM steps of ``attack_mixed`` NO sample takes a feature step,
    the mixed closure returns no feature list at all and the step is the cross entropy alone.  The result must equal the
    schedule-pure bucket path (``attack_batch(dual=True, tasks=...)``) and each sample's batch-1 oracle loop."""
    from oracle import text_scoring as ts
    from vqattack_amd.attack import mlm_task
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build(flavor)
    g = torch.Generator().manual_seed(71)
    images = torch.empty(2, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    ids = IDS[:2]
    masks = (ids != 0).long()
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[0, 1] = att[1, 2] = True                                   # one substitutable word each: blocks [4, 8] + 1 probe
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    max_len = cfg.max_text_len if flavor == "vlmo" else None
    tasks, oracle_tasks = [], []
    for s in range(2):
        body = [(int(t),) for t in ids[s].tolist() if t not in (0, 101, 102)]
        answer = (7301 + s,)
        para = body[:2] + [answer]
        correct = [[answer]] + ([[(7401,)]] if s else [])
        same = [True] + [False] * (len(correct) - 1)
        tasks.append(mlm_task.build_mlm_task([answer], correct, same, para, [], flavor, max_len=max_len))
        ot = ts.build_mlm_task([answer], correct, same, para, [], flavor)
        if flavor == "vlmo":
            ot["text_ids_mlm"], ot["text_mask_mlm"] = ts.encode_words(ot["list_words"], max_len, max_len)
            lab = ot["mlm_labels"]
            ot["mlm_labels"] = [row[:max_len] for row in lab] if isinstance(lab[0], list) else lab[:max_len]
        ot["tail"] = ()
        oracle_tasks.append(ot)
    attack = BatchedVQAttack(adapters_cls(gpu_model), flavor, gpu_model.embedding_tables(),
                             AttackConfig(budget=12, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
    proposals = text_update.propose_candidates(attack.adapters.mlm_logits(ids.to(DEV), masks.to(DEV)), ids, att,
                                               threshold=0)
    args = (images.to(DEV), ids.to(DEV), masks.to(DEV), att.to(DEV))
    mixed = attack.attack_mixed(*args, init_eta=eta.to(DEV), proposals=proposals, tasks=tasks)
    bucket = attack.attack_batch(*args, init_eta=eta.to(DEV), proposals=proposals, dual=True, tasks=tasks)
    assert mixed.gradient_steps == 2 * 13 and bucket.gradient_steps == 13
    assert torch.equal(mixed.adv_text_ids, bucket.adv_text_ids)
    assert (mixed.adv_images == bucket.adv_images).float().mean().item() >= 0.999
    for s in range(2):
        n = int(masks[s].sum()) if flavor == "albef" else ids.shape[1]
        if flavor == "vlmo":
            orig = ts.encode_words
            ts.encode_words = lambda words, _ml, _pad, tail=(): orig(words, max_len, max_len, tail)
        try:
            adv, new_ids, _ = attack_loop.attack_one(ref_cls, cpu_model, flavor, images[s:s + 1], ids[s:s + 1, :n],
                                                     masks[s:s + 1, :n], proposals[s], sim, init_eta=eta[s:s + 1],
                                                     budget=12, sim_threshold=0.3, task=oracle_tasks[s])
        finally:
            if flavor == "vlmo":
                ts.encode_words = orig
        assert mixed.adv_text_ids[s, :n].cpu().tolist() == new_ids[0].tolist(), (flavor, s)
        assert (mixed.adv_images[s].cpu() == adv[0]).float().mean().item() >= 0.99, (flavor, s)


def test_vlmo_mixed_loss_on_a_ragged_batch_matches_per_sample_oracle():
    """The VLMO copy's third loss (``ls`` not in {0, 1}: feature loss / (layers * Ntok) + 0.1 * CE(labels) + 0.1 * sum of
    CE(synonym label sets), ``VLMO_VQAttack/cleverhans/.../fast_gradient_method.py:127-131``; not reached by the
    reference's drivers) on a BATCH of three questions of different lengths: every sample is weighted by its own token
    count and its own label counts, so each sample's FGM step equals the batch-1 oracle's, and the reported loss is the
    sum of the three batch-1 losses."""
    import numpy as np
    from oracle import cleverhans_cpu as o
    from vqattack_amd import attacks
    cpu_model, gpu_model, adapters_cls, ref_cls, cfg = _build("vlmo")
    g = torch.Generator().manual_seed(51)
    images = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    start = torch.clamp(images + torch.empty_like(images).uniform_(-0.1, 0.1, generator=g), -1, 1)
    masks = (IDS != 0).long()
    labels = torch.full_like(IDS, -100)
    labels[0, 2], labels[1, 3], labels[1, 4], labels[2, 1] = 2003, 2024, 2051, 2054
    syn = [torch.full_like(IDS, -100), torch.full_like(IDS, -100)]
    syn[0][0, 2], syn[0][1, 3], syn[0][2, 1] = 2004, 2025, 2055
    syn[1][0, 3], syn[1][1, 4], syn[1][2, 2] = 2061, 2052, 2060   # (a set without any label of a sample is NaN in the
    #                                                               reference's batch-1 F.cross_entropy: not a case)
    ad = adapters_cls(gpu_model)
    ad.set_text(IDS.to(DEV), masks.to(DEV))
    ad.set_mlm_rows(None)
    targets = ad.gen_ori_feats(images.to(DEV))
    y = [labels.to(DEV), targets[1], targets[2], [[s.to(DEV)] for s in syn]]
    adv, loss = attacks.fast_gradient_method(ad.pgd_mlm_attack, start.to(DEV), 0.01, np.inf, images.to(DEV), clip_min=-1,
                                             clip_max=1, y=y, ls=2, flavor="vlmo", sanity_checks=True)
    want_loss = 0.0
    for s in range(3):
        ref = ref_cls(cpu_model, IDS[s:s + 1], masks[s:s + 1])
        t = ref.gen_ori_feats(images[s:s + 1])
        ys = [labels[s:s + 1], t[1], t[2], [[x[s:s + 1]] for x in syn]]
        with torch.enable_grad():
            want, ls_s = o.fast_gradient_method(ref.pgd_mlm_attack, start[s:s + 1], 0.01, np.inf, images[s:s + 1],
                                                clip_min=-1, clip_max=1, y=ys, ls=2, flavor="vlmo")
        want_loss += float(ls_s.detach()) if torch.is_tensor(ls_s) else float(ls_s)
        same = (adv[s].cpu() == want[0].detach()).float().mean().item()
        assert same >= 0.99, (s, same)
    assert abs(float(loss) - want_loss) <= 2e-4 * abs(want_loss)
