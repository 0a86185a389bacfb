"""Batched VQAttack driver: the block loop of the reference's orchestrators on top of the drop-in operators.

Reference control flow (one sample at a time, batch 1): ``Adv_attack.evaluate``, ``ALBEF_attack/adv_attack.py:560-716``
and ``VLMo.test_step``, ``vlmo/modules/vlmo_module.py:1893-2062``:

    targets  = Gen_ori_feats(clean image, clean question)
    blocks   = cal_text_attack_list(question)                      # schedule.iter_schedule + MLM candidates
    no substitutable word:  one PGD of 40 steps (20 dual-loss iterations)
    else for each block i:  re-tokenise adv text; PGD(nb_iter = blocks[i], time = 0 only for i == 0)
                            unless last: pgd_vl(1 step, attack_mask) -> update_adv_text
    score (adv image, adv text) with the black box; success = answer differs from the clean answer

Here the same flow runs for a whole batch whose samples share a schedule (``schedule.bucket_by_schedule``); the
operators are the shipped HIP path (``vqattack_amd.dropin``), the white box is any adapter object exposing the
reference's closures (``pgd_attack``, ``pgd_attack_vl``, ``pgd_mlm_attack``, ``gen_ori_feats``, ``set_text``).
Attack hyper-parameters default to the reference's hard-coded literals (adv_attack.py:607-610: eps 0.125, step 0.01,
L-inf, clip [-1, 1]).
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from .. import attacks, dropin, layout, ops
from ..features import LayerFeatures
from . import mlm_task, text_update
from .schedule import IMAGE_STEP_BUDGET, iter_schedule


@dataclass
class AttackConfig:
    eps: float = 0.125
    eps_iter: float = 0.01
    budget: int = IMAGE_STEP_BUDGET
    clip_min: float = -1.0
    clip_max: float = 1.0
    norm: float = np.inf
    random_start: bool = True          # first block uses time=0 (uniform start), like the reference
    sanity_checks: bool = False        # the flag read is a host sync; parity tests switch it on
    sim_threshold: float = text_update.SIM_THRESHOLD   # adv_attack.py:303
    use_graph: bool = False            # replay PGD iterations from a hipGraph (small, launch-bound batches)
    patch_layout: bool = False         # keep the PGD state patch-major (layout.py); measured neutral end to end, see DESIGN


@dataclass
class BatchResult:
    adv_images: torch.Tensor
    adv_text_ids: torch.Tensor
    loss_lists: list = field(default_factory=list)
    substitutions: list = field(default_factory=list)   # per round: per sample [(position, old id, new id), ...]
    gradient_steps: int = 0
    adv_text_ids_mlm: torch.Tensor = None               # dual-loss samples: the [MASK]-ed paraphrase after the attack


def stack_mlm_tasks(tasks, device, dtype=torch.int64):
    """(ids (B, Lm), mask (B, Lm), labels (B, Lm) or (B, K, Lm)) of a batch of ``mlm_task.MlmTask`` with old_alg == 0.
    Samples are padded to the longest paraphrase (pad id 0, mask 0, label -100) and to the largest number of label sets
    (an all-ignored set adds nothing to a sample's loss or gradient)."""
    lm = max(len(t.text_ids_mlm) for t in tasks)
    sets = [t.mlm_labels if isinstance(t.mlm_labels[0], list) else [t.mlm_labels] for t in tasks]
    k = max(len(x) for x in sets)
    ids = torch.zeros(len(tasks), lm, dtype=dtype)
    mask = torch.zeros(len(tasks), lm, dtype=dtype)
    labels = torch.full((len(tasks), k, lm), mlm_task.IGNORE, dtype=torch.int64)
    for b, (t, ls) in enumerate(zip(tasks, sets)):
        n = len(t.text_ids_mlm)
        ids[b, :n] = torch.tensor(t.text_ids_mlm)
        mask[b, :n] = torch.tensor(t.text_mask_mlm)
        for j, row in enumerate(ls):
            labels[b, j, :len(row)] = torch.tensor(row)
    if k == 1:
        labels = labels[:, 0]
    return ids.to(device), mask.to(device), labels.to(device)


class BatchedVQAttack:
    def __init__(self, adapters, flavor, tables, config=None, similarity_fn=None, banned_ids=None):
        self.adapters = adapters
        self.flavor = flavor
        self.tables = tables                     # embedding tables for the candidate-scoring kernel
        self.cfg = config or AttackConfig()
        self.similarity_fn = similarity_fn or text_update.BagOfEmbeddingsSimilarity()
        self.banned_ids = banned_ids
        ns = dropin.load(flavor)
        self.pgd = ns.projected_gradient_descent.projected_gradient_descent
        self.pgd_vl = ns.projected_gradient_descent_vl.projected_gradient_descent

    # the y lists the reference passes (adv_attack.py:609,637; vlmo_module.py:1948,1975)
    def _y_feature(self, targets):
        if self.flavor == "albef":
            return [targets[0], targets[1], None, None, None]
        return [targets[0], targets[1], targets[2]]

    def _y_dual(self, targets, mlm_labels):
        if self.flavor == "albef":
            return [mlm_labels, targets[0], targets[1]]
        return [mlm_labels, targets[1], targets[2]]

    def _pgd_block(self, adv, clean, targets, steps, time, dual, mlm_labels, init_eta=None):
        c = self.cfg
        common = dict(clip_min=c.clip_min, clip_max=c.clip_max, time=time, ori_x=clean,
                      sanity_checks=c.sanity_checks, init_eta=init_eta)
        a = self.adapters
        if not dual:
            return self.pgd(a.pgd_attack, adv, c.eps, c.eps_iter, steps, c.norm, y=self._y_feature(targets), ls=1,
                            graph=c.use_graph and steps > 2, **common)
        return self.pgd([a.pgd_attack, a.pgd_mlm_attack], adv, c.eps, c.eps_iter, steps // 2, c.norm,
                        y=self._y_dual(targets, mlm_labels), ls=0, per_sample=True, **common)

    @torch.no_grad()
    def attack_batch(self, images, text_ids, text_masks, attackable, mlm_logits_fn=None, dual=False,
                     mlm_labels=None, init_eta=None, proposals=None, text_ids_mlm=None, text_mask_mlm=None, tasks=None):
        """Attack one batch whose samples all have the same number of attackable words.

        images (B,3,H,W) in [clip_min, clip_max]; text_ids/text_masks (B,L); attackable bool (B,L) with the same
        count per row.  ``mlm_logits_fn(text_ids, text_masks) -> (B,L,V)`` proposes substitution candidates (the
        reference uses a separate HF BERT-MLM, adv_attack.py:110,242; default: the adapters' own ``mlm_logits``);
        ``proposals`` injects them directly (``text_update.propose_candidates`` format).  Returns a ``BatchResult``.

        ``dual=True`` runs the reference's ``old_alg == 0`` blocks (feature step + MLM step per iteration,
        adv_attack.py:614-619,670-676).  The MLM side is given either as ``tasks`` -- one ``mlm_task.MlmTask`` per sample
        (``mlm_task.build_mlm_task``): ids, masks and labels are stacked from them and every accepted substitution is
        applied to the paraphrase's word list like ``update_mlm_text`` (:334-351) -- or as ready tensors
        ``mlm_labels`` (B, L) / (B, K, L), ``text_ids_mlm``, ``text_mask_mlm`` that stay fixed during the attack.
        """
        c, a = self.cfg, self.adapters
        dev = images.device
        n_words = int(attackable[0].sum().item())
        if not bool((attackable.sum(dim=1) == n_words).all()):
            raise ValueError("samples of one batch must share a schedule: bucket them with bucket_by_schedule()")
        blocks = iter_schedule(n_words, c.budget)
        images, init_eta, restore = self._enter_layout(images, init_eta)
        adv = images
        adv_ids = text_ids.clone().contiguous()
        mlm_ids = mlm_mask = None
        if tasks is not None:
            if not dual or any(t.old_alg != 0 for t in tasks):
                raise ValueError("tasks= is for dual-loss batches: every MlmTask must have old_alg == 0")
            tasks = [mlm_task.MlmTask(**vars(t)) for t in tasks]                 # private copies: word lists are edited
            for t in tasks:
                t.words_mlm = list(t.words_mlm)
            mlm_ids, mlm_mask, mlm_labels = stack_mlm_tasks(tasks, dev, text_ids.dtype)
        elif text_ids_mlm is not None:
            mlm_ids = text_ids_mlm.clone()
            mlm_mask = text_mask_mlm
        # the text layout (trimmed padding) is fixed here for targets and every later step: question AND paraphrase masks
        a.set_text(text_ids, text_masks, text_ids_mlm=mlm_ids, text_mask_mlm=mlm_mask)
        # adapters that trim padding zero-pad the MLM logits back to the full text length (VlmoAttackAdapters): a label
        # there would be scored against those zeros
        tlen = getattr(a, "_tlen", None) if getattr(a, "trim_padding", False) else None
        if dual and mlm_labels is not None and tlen is not None and tlen < mlm_labels.shape[-1] and \
                bool((mlm_labels[..., tlen:] != mlm_task.IGNORE).any()):
            raise ValueError("MLM labels beyond the batch's text length {} must be ignore_index".format(tlen))
        targets = a.gen_ori_feats(images)
        res = BatchResult(adv_images=adv, adv_text_ids=adv_ids)
        first_time = 0 if c.random_start else 1
        rounds = []
        if not blocks:
            with torch.enable_grad():
                adv, losses = self._pgd_block(adv, images, targets, c.budget, first_time, dual, mlm_labels, init_eta)
            res.loss_lists.append(losses)
            res.gradient_steps = c.budget
        else:
            if proposals is None:
                fn = mlm_logits_fn or getattr(a, "mlm_logits", None)
                if fn is not None:
                    proposals = text_update.propose_candidates(fn(text_ids, text_masks), text_ids, attackable,
                                                               banned=self.banned_ids)
            plan = None if proposals is None else text_update.CandidatePlan(proposals, dev)   # uploaded once
            # text embeddings of the question batch: one launch; after every substitution round the batch is re-embedded
            # from the device-resident ids (masked-token embedding substitution, ops.embed_tokens) -- the host never has
            # to learn which words changed
            e_ori = ops.embed_tokens(self.tables, text_ids)
            adv_emb = e_ori.clone()
            ori_ids = text_ids.contiguous()
            positions = ops.RowIndex(range(text_ids.shape[1]), text_ids.shape[1], dev)   # probe mask, uploaded once
            for bi, steps in enumerate(blocks):
                a.set_text(adv_ids, text_masks, text_ids_mlm=mlm_ids, text_mask_mlm=mlm_mask)
                with torch.enable_grad():
                    adv, losses = self._pgd_block(adv, images, targets, steps, first_time if bi == 0 else 1, dual,
                                                  mlm_labels, init_eta if bi == 0 else None)
                res.loss_lists.append(losses)
                res.gradient_steps += steps
                if bi == len(blocks) - 1:
                    break
                with torch.enable_grad():
                    adv, text_grad = self.pgd_vl(a.pgd_attack_vl, [adv, adv_emb], c.eps, c.eps_iter,
                                                 1, c.norm, clip_min=c.clip_min, clip_max=c.clip_max,
                                                 y=self._y_feature(targets), time=1, ori_x=images, ls=1,
                                                 attack_mask=positions, sanity_checks=c.sanity_checks)
                res.gradient_steps += 1
                if plan is not None:
                    scores = text_update.score_plan(self.tables, e_ori, text_grad, plan)
                    prev = adv_ids.clone()
                    new_id, rank = text_update.accept_round(plan, scores, ori_ids, adv_ids, self.similarity_fn,
                                                            c.sim_threshold)            # adv_ids updated in place
                    rounds.append((prev, new_id, rank))
                    ops.embed_tokens(self.tables, adv_ids, out=adv_emb)
                    if tasks is not None:        # update_mlm_text: the paraphrase follows the question's substitutions
                        subs = text_update.substitution_lists(prev, new_id, rank)
                        for s, per in enumerate(subs):
                            if per:
                                mlm_task.apply_substitutions(tasks[s].words_mlm, [(old, new) for (_, old, new) in per])
                                ids_s = tasks[s].reencode()
                                mlm_ids[s, :len(ids_s)] = torch.tensor(ids_s, device=dev, dtype=mlm_ids.dtype)
        res.substitutions = [text_update.substitution_lists(*r) for r in rounds]     # one host read, after the attack
        res.adv_images, res.adv_text_ids, res.adv_text_ids_mlm = restore(adv), adv_ids, mlm_ids
        return res

    def _enter_layout(self, images, init_eta):
        """Switch the PGD state to patch-major layout when enabled and supported by the white box (its config names a
        patch size); returns (images, init_eta, restore) with ``restore`` mapping the result back to (B, 3, H, W)."""
        patch = getattr(getattr(getattr(self.adapters, "model", None), "cfg", None), "patch", None)
        if not self.cfg.patch_layout or patch is None or images.dim() != 4:
            return images, init_eta, (lambda t: t)
        _, ch, h, w = images.shape
        eta = None if init_eta is None else layout.to_patches(init_eta, patch)
        return layout.to_patches(images, patch), eta, (lambda t: layout.from_patches(t, patch, h, w, ch))

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def attack_mixed(self, images, text_ids, text_masks, attackable, mlm_logits_fn=None, init_eta=None,
                     proposals=None):
        """Feature-loss joint attack of a batch whose samples have DIFFERENT numbers of attackable words.

        A sample with ``w`` substitutable words is, end to end, one sequence of ``budget + w`` L-inf steps whose text
        changes after each of its probe steps (the reference's per-block ``projected_gradient_descent`` calls restart
        from a feasible point with eta = 0, which is a no-op).  So the batch runs ``max(budget + w)`` global steps;
        samples are sorted by length so that the still-active ones are always a batch PREFIX (finished samples drop out
        of the white-box batch instead of being masked), every step takes the image+text-embedding gradient, and a
        sample's word substitution fires right after its own probe steps.  Per-sample results equal ``attack_batch`` on
        schedule-pure buckets (tests/test_attack_batched_parity.py).
        """
        c, a = self.cfg, self.adapters
        if c.norm != np.inf:
            raise ValueError("attack_mixed implements the L-inf feature-loss attack")
        dev, b = images.device, images.shape[0]
        images, init_eta, restore = self._enter_layout(images, init_eta)
        n_words = attackable.sum(dim=1).tolist()
        total = [c.budget + int(w) for w in n_words]
        order = sorted(range(b), key=lambda s: -total[s])                 # longest first: active set = prefix
        perm = torch.tensor(order, device=dev)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(b, device=dev)
        images, text_ids, text_masks, attackable = images[perm], text_ids[perm], text_masks[perm], attackable[perm]
        if init_eta is not None:
            init_eta = init_eta[perm]
        total = [total[s] for s in order]
        probes = []                                                       # per sample: global step indices of its probes
        for s in order:
            blocks, at, mine = iter_schedule(int(n_words[s]), c.budget), 0, set()
            for j, blen in enumerate(blocks[:-1]):
                at += blen
                mine.add(at + j)
            probes.append(mine)
        if proposals is None:
            fn = mlm_logits_fn or getattr(a, "mlm_logits", None)
            proposals = text_update.propose_candidates(fn(text_ids, text_masks), text_ids, attackable,
                                                       banned=self.banned_ids) if fn is not None else [[] for _ in range(b)]
        else:
            proposals = [proposals[s] for s in order]
        plan = text_update.CandidatePlan(proposals, dev)
        a.set_text(text_ids, text_masks)
        pinned = getattr(a, "_tlen", None)                                # token layout of the targets
        targets = a.gen_ori_feats(images)
        e_ori = ops.embed_tokens(self.tables, text_ids)
        adv_emb = e_ori.clone()
        adv_ids = text_ids.clone().contiguous()
        ori_ids = text_ids.contiguous()
        rounds = []
        flag = ops.new_flag(dev)
        eta = init_eta
        if eta is None and c.random_start:
            eta = torch.empty_like(images).uniform_(-c.eps, c.eps)
        cur = ops.linf_init(images.contiguous(), eta, c.eps, c.clip_min, c.clip_max, flag=flag)
        losses = torch.zeros(max(total), dtype=torch.float32, device=dev)
        ws = ops.Workspace()
        res = BatchResult(adv_images=cur, adv_text_ids=adv_ids)
        n_act_prev = None
        for t in range(max(total)):
            n_act = sum(1 for x in total if x > t)
            if n_act != n_act_prev:
                a.set_text(adv_ids[:n_act], text_masks[:n_act], text_len=pinned)
                y = [v.rows(n_act) if isinstance(v, LayerFeatures) else (None if v is None else v[:n_act])
                     for v in self._y_feature(targets)]
                n_act_prev = n_act
            leaf_img = cur[:n_act].detach().requires_grad_(True)
            leaf_txt = adv_emb[:n_act].detach().requires_grad_(True)
            attacks._loss_and_grad(a.pgd_attack_vl, [leaf_img, leaf_txt], [leaf_img, leaf_txt], list(y), 1, self.flavor,
                                   False, attacks._LossSlot(losses, t), vl=True, ws=ws)
            ops.linf_step(cur[:n_act], leaf_img.grad, images[:n_act], c.eps_iter, c.eps, c.clip_min, c.clip_max,
                          out=cur[:n_act])                                # in place: finished samples stay untouched
            firing = [s for s in range(n_act) if t in probes[s]]
            if firing:
                sub = plan.restricted_to(firing)
                scores = text_update.score_plan(self.tables, e_ori[:n_act], leaf_txt.grad, sub)
                prev = adv_ids[:n_act].clone()
                new_id, rank = text_update.accept_round(sub, scores, ori_ids[:n_act], adv_ids[:n_act],
                                                        self.similarity_fn, c.sim_threshold)   # in place, on the device
                rounds.append((prev, new_id, rank))
                ops.embed_tokens(self.tables, adv_ids, out=adv_emb)
                a.set_text(adv_ids[:n_act], text_masks[:n_act], text_len=pinned)
        if c.sanity_checks:
            assert int(flag.item()) == 0, "input images are outside [clip_min, clip_max]"
        res.substitutions = [text_update.substitution_lists(*r) for r in rounds]
        res.adv_images, res.adv_text_ids = restore(cur[inv]), adv_ids[inv]
        res.loss_lists = [losses.tolist()]
        res.gradient_steps = sum(total)
        return res
