"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the per-sample block loop of the joint attack (batch 1, like the reference).

Restated from ``Adv_attack.evaluate`` (``ALBEF_attack/adv_attack.py:559-712``) / ``VLMo.test_step``
(``vlmo_module.py:1892-2057``) on token ids: PGD blocks through the pinned oracle operators, one ``pgd_vl`` probe step
between blocks, then ``update_adv_text`` (:265-324) and, on the dual-loss path (``old_alg == 0``), ``update_mlm_text``
(:334-351).

**Pinned** by the ``loop_albef`` / ``loop_vlmo`` cases of ``tests/golden/text_golden.*``: the reference's own loop code
(compiled from its source and executed with its own cleverhans operators over this repository's tiny white boxes,
``tests/golden/make_text_golden_tasks.py``) -- ``tests/test_text_golden_loops.py`` compares adversarial images, token ids
and per-block loss lists.
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
               eps=0.125, eps_iter=0.01, budget=40, sim_threshold=0.95, task=None):
    """One (image (1,3,H,W), question (1,L)) pair.  ``proposals``: [(position, [ids...]), ...] for this sample, in word
    order.  ``adapters_factory(model, ids, masks, ids_mlm=None, masks_mlm=None)`` builds reference-style batch-1
    closures (oracle/adapters_ref.py).  ``task``: result of ``text_scoring.build_mlm_task`` with ``old_alg == 0`` ->
    the dual-loss blocks (feature step + MLM step per iteration, ``int(iter / 2)`` iterations per block).
    Returns ``(adv image, adv ids (1, L), [loss list per PGD call])``."""
    n_words = len(proposals) if proposals is not None else 0
    blocks = ts.iter_schedule(n_words, budget)
    dual = task is not None and task["old_alg"] == 0
    ids_mlm = masks_mlm = labels = list_words = None
    if dual:
        ids_mlm = torch.tensor([task["text_ids_mlm"]], dtype=text_ids.dtype)
        masks_mlm = torch.tensor([task["text_mask_mlm"]], dtype=text_masks.dtype)
        lab = task["mlm_labels"]
        labels = torch.tensor([lab], dtype=torch.long)          # (1, L) or (1, K, L)
        list_words = list(task["list_words"])
    ad = adapters_factory(model, text_ids, text_masks, ids_mlm, masks_mlm)
    targets = ad.gen_ori_feats(image)

    def y_feat():
        return [targets[0], targets[1], None, None, None] if flavor == "albef" else list(targets)

    def y_dual():
        return [labels, targets[0], targets[1]] if flavor == "albef" else [labels, targets[1], targets[2], None]

    def pgd_block(ad, adv, steps, time, eta):
        common = dict(clip_min=-1, clip_max=1, ori_x=image, time=time, flavor=flavor, init_eta=eta)
        with torch.enable_grad():
            if not dual:
                return o.projected_gradient_descent(ad.pgd_attack, adv, eps, eps_iter, steps, np.inf, y=y_feat(), ls=1,
                                                    **common)
            return o.projected_gradient_descent([ad.pgd_attack, ad.pgd_mlm_attack], adv, eps, eps_iter, int(steps / 2),
                                                np.inf, y=y_dual(), ls=0, **common)

    adv, ids = image, text_ids.clone()
    losses = []
    tables = model.embedding_tables()
    tab = (tables["word"], tables["pos"], tables["type_emb"], tables["gamma"], tables["beta"], tables["ln_eps"])
    if not blocks:
        adv, ll = pgd_block(ad, adv, budget, 0, init_eta)
        return adv, ids, [ll]
    e_ori = ts.bert_embeddings(text_ids, *tab)
    attack_vector = [p for (p, _) in proposals]
    cand_ids = [vs for (_, vs) in proposals]
    sub_list = list(range(len(proposals)))
    for bi, steps in enumerate(blocks):
        ad = adapters_factory(model, ids, text_masks, ids_mlm, masks_mlm)
        adv, ll = pgd_block(ad, adv, steps, 0 if bi == 0 else 1, init_eta if bi == 0 else None)
        losses.append(ll)
        if bi == len(blocks) - 1:
            break
        with torch.enable_grad():
            emb = ts.bert_embeddings(ids, *tab)
            adv, tgrad = o.projected_gradient_descent_vl(ad.pgd_attack_vl, [adv, emb], eps, eps_iter, 1, np.inf,
                                                         clip_min=-1, clip_max=1, y=y_feat(), ori_x=image, time=1,
                                                         ls=1, attack_mask=attack_vector, flavor=flavor)
        new, ops = ts.update_adv_text(tgrad, cand_ids, sub_list, attack_vector, ids[0].tolist(), text_ids[0].tolist(),
                                      e_ori, tab, similarity_fn, sim_threshold)
        ids = torch.tensor([new], dtype=text_ids.dtype)
        if dual:
            ts.update_mlm_text(ops, list_words)
            max_len, pad_to = (25, None) if flavor == "albef" else (40, 40)
            enc, msk = ts.encode_words(list_words, max_len, pad_to, tail=task.get("tail", ()))
            ids_mlm = torch.tensor([enc], dtype=text_ids.dtype)
            masks_mlm = torch.tensor([msk], dtype=text_masks.dtype)
    return adv, ids, losses
