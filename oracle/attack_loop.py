"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the per-sample block loop of the joint attack (batch 1, like the reference).

Restated from ``Adv_attack.evaluate`` (``ALBEF_attack/adv_attack.py:604-712``) / ``VLMo.test_step``
(``vlmo_module.py:1943-2055``) on token ids: PGD blocks through the pinned oracle operators, one ``pgd_vl`` probe
step between blocks, then ``update_adv_text`` (:265-324).  The orchestrators cannot be imported in the build container
(tensorflow_hub, timm, sacred, checkpoints), so this loop is **parity unpinned**; the operators it calls are pinned.
"""
import numpy as np
import torch

from . import cleverhans_cpu as o
from . import text_scoring as ts


def greedy_accept_one(cand_rows, scores, ori_ids, cur_ids, similarity_fn, threshold=0.95):
    """update_adv_text :300-323 for one sample.  cand_rows: list of (position, vocabulary id)."""
    order = sorted(range(len(cand_rows)), key=lambda k: scores[k], reverse=True)
    new_ids = list(cur_ids)
    taken = set()
    for k in order:
        p, v = cand_rows[k]
        if p in taken:
            continue
        trial = list(new_ids)
        trial[p] = v
        sim = float(similarity_fn([int(t) for t in ori_ids], [int(t) for t in trial]))
        if sim > threshold:
            threshold = sim
            taken.add(p)
            new_ids = trial
    return new_ids


def attack_one(adapters_factory, model, flavor, image, text_ids, text_masks, proposals, similarity_fn, init_eta=None,
               eps=0.125, eps_iter=0.01, budget=40, sim_threshold=0.95):
    """One (image (1,3,H,W), question (1,L)) pair.  ``proposals``: [(position, [ids...]), ...] for this sample.
    ``adapters_factory(model, ids, masks)`` builds reference-style batch-1 closures (oracle/adapters_ref.py)."""
    n_words = len(proposals) if proposals is not None else 0
    blocks = ts.iter_schedule(n_words, budget)
    ad = adapters_factory(model, text_ids, text_masks)
    targets = ad.gen_ori_feats(image)

    def y_feat():
        return [targets[0], targets[1], None, None, None] if flavor == "albef" else list(targets)

    adv, ids = image, text_ids.clone()
    losses = []
    tables = model.embedding_tables()
    tab = (tables["word"], tables["pos"], tables["type_emb"], tables["gamma"], tables["beta"], tables["ln_eps"])
    if not blocks:
        with torch.enable_grad():
            adv, ll = o.projected_gradient_descent(ad.pgd_attack, adv, eps, eps_iter, budget, np.inf, clip_min=-1,
                                                   clip_max=1, y=y_feat(), ori_x=image, time=0, ls=1, flavor=flavor,
                                                   init_eta=init_eta)
        return adv, ids, [ll]
    e_ori = ts.bert_embeddings(text_ids, *tab)
    positions = list(range(text_ids.shape[1]))
    for bi, steps in enumerate(blocks):
        ad = adapters_factory(model, ids, text_masks)
        with torch.enable_grad():
            adv, ll = o.projected_gradient_descent(ad.pgd_attack, adv, eps, eps_iter, steps, np.inf, clip_min=-1,
                                                   clip_max=1, y=y_feat(), ori_x=image, time=0 if bi == 0 else 1,
                                                   ls=1, flavor=flavor, init_eta=init_eta if bi == 0 else None)
        losses.append(ll)
        if bi == len(blocks) - 1:
            break
        with torch.enable_grad():
            emb = ts.bert_embeddings(ids, *tab)
            adv, tgrad = o.projected_gradient_descent_vl(ad.pgd_attack_vl, [adv, emb], eps, eps_iter, 1, np.inf,
                                                         clip_min=-1, clip_max=1, y=y_feat(), ori_x=image, time=1,
                                                         ls=1, attack_mask=positions, flavor=flavor)
        rows = [(p, v) for (p, vs) in proposals for v in vs]
        cand = torch.tensor([[0, p, p, v] for (p, v) in rows], dtype=torch.int64)
        scores = ts.candidate_scores(text_ids, e_ori, tgrad, cand, *tab).tolist() if rows else []
        new = greedy_accept_one(rows, scores, text_ids[0].tolist(), ids[0].tolist(), similarity_fn, sim_threshold)
        ids = torch.tensor([new], dtype=text_ids.dtype)
    return adv, ids, losses
