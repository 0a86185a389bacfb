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
import torch.nn.functional as F

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
    live_mlm_rows: bool = True         # dual loss: MLM head + cross entropy at the live label rows only (False: dense B x L)


@dataclass
class BatchResult:
    adv_images: torch.Tensor
    adv_text_ids: torch.Tensor
    loss_lists: list = field(default_factory=list)
    substitutions: list = field(default_factory=list)   # per round: per sample (CALLER's order, one entry per sample of
    #                                                      the batch) [(position, old id, new id), ...]
    gradient_steps: int = 0      # attack_batch: steps of ONE sample (= white-box passes); attack_mixed: sum over samples
    global_steps: int = 0        # white-box forward + backward passes the call ran (both drivers)
    sample_steps: int = 0        # sum over the batch's samples of each sample's own gradient steps (both drivers)
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
        try:
            return self._attack_batch_body(images, text_ids, text_masks, attackable, mlm_logits_fn, dual, mlm_labels,
                                           init_eta, proposals, tasks, mlm_ids, mlm_mask, blocks, adv, adv_ids, restore)
        finally:                                 # also after an exception (OOM, bad label, HIP error): the adapters' MLM
            if hasattr(a, "set_mlm_rows"):       # closure is dense again for any other caller
                a.set_mlm_rows(None)

    def _attack_batch_body(self, images, text_ids, text_masks, attackable, mlm_logits_fn, dual, mlm_labels, init_eta,
                           proposals, tasks, mlm_ids, mlm_mask, blocks, adv, adv_ids, restore):
        c, a = self.cfg, self.adapters
        dev = images.device
        if hasattr(a, "set_mlm_rows"):
            # live-rows form of the MLM closure: the head runs, and the cross entropy is taken, at the label positions that
            # are targets only (the [MASK]-ed answer pieces, adv_attack.py:433-558) -- not over all B x L positions
            if dual and mlm_labels is not None and c.live_mlm_rows:
                live_rows, mlm_labels = mlm_task.live_label_rows(mlm_labels)
                a.set_mlm_rows(live_rows)
            else:
                a.set_mlm_rows(None)
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
        res.global_steps, res.sample_steps = res.gradient_steps, res.gradient_steps * images.shape[0]
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
    @staticmethod
    def _step_kinds(n_words, budget, dual):
        """A sample's own sequence of white-box gradient steps as (kind, is_probe) pairs.  Kinds: ``F`` feature-loss step
        followed by the eps-ball projection (every step of an ``old_alg == 1`` sample, and every probe); ``N`` feature
        step WITHOUT projection and ``M`` MLM step with projection -- the two halves of one dual-loss iteration
        (projected_gradient_descent.py:153-189), ``int(iter / 2)`` iterations per block (adv_attack.py:614-619,670-676)."""
        blocks = iter_schedule(n_words, budget) or [budget]
        seq = []
        for j, blen in enumerate(blocks):
            seq += [("N", False), ("M", False)] * (blen // 2) if dual else [("F", False)] * blen
            if j < len(blocks) - 1:
                seq.append(("F", True))
        return seq

    @torch.no_grad()
    def attack_mixed(self, images, text_ids, text_masks, attackable, mlm_logits_fn=None, init_eta=None,
                     proposals=None, tasks=None):
        """Joint attack of a batch whose samples have DIFFERENT numbers of attackable words and DIFFERENT loss modes.

        A sample with ``w`` substitutable words is, end to end, one sequence of ``budget + w`` L-inf steps whose text
        changes after each of its probe steps (the reference's per-block ``projected_gradient_descent`` calls restart
        from a feasible point with eta = 0, which is a no-op).  So the batch runs ``max(budget + w)`` global steps;
        samples are sorted by length so that the still-active ones are always a batch PREFIX (finished samples drop out
        of the white-box batch instead of being masked), every step takes the image+text-embedding gradient, and a
        sample's word substitution fires right after its own probe steps.  Per-sample results equal ``attack_batch`` on
        schedule-pure buckets (tests/test_attack_batched_parity.py).

        ``tasks`` (optional, one ``mlm_task.MlmTask`` or None per sample): samples whose task has ``old_alg == 0`` run the
        reference's dual-loss blocks -- their steps alternate between a feature step without projection and an MLM step
        on the [MASK]-ed paraphrase (``_step_kinds``).  All samples still share ONE white-box pass per global step
        (``adapters.pgd_attack_mixed``): a sample at an MLM step is run with its paraphrase as text, its feature rows
        weigh 0 and its MLM head is evaluated at its live label rows only; the fused image update is launched per run of
        consecutive samples of one kind.  The paraphrase follows the question's substitutions (``update_mlm_text``).
        """
        a = self.adapters
        try:
            return self._attack_mixed_body(images, text_ids, text_masks, attackable, mlm_logits_fn, init_eta, proposals,
                                           tasks)
        finally:                                 # also after an exception: no live-rows / MLM-sample state may linger
            if hasattr(a, "set_mlm_rows"):
                a.set_mlm_rows(None)
            if hasattr(a, "set_mlm_samples"):
                a.set_mlm_samples(None)

    def _attack_mixed_body(self, images, text_ids, text_masks, attackable, mlm_logits_fn, init_eta, proposals, tasks):
        c, a = self.cfg, self.adapters
        if c.norm != np.inf:
            raise ValueError("attack_mixed implements the L-inf attack")
        dev, b, text_len = images.device, images.shape[0], text_ids.shape[1]
        images, init_eta, restore = self._enter_layout(images, init_eta)
        n_words = [int(w) for w in attackable.sum(dim=1).tolist()]
        is_dual = [tasks is not None and tasks[s] is not None and tasks[s].old_alg == 0 for s in range(b)]
        kinds = [self._step_kinds(n_words[s], c.budget, is_dual[s]) for s in range(b)]
        # longest first: the active set is a prefix; equal lengths keep the loss modes together (fewer update launches)
        order = sorted(range(b), key=lambda s: (-len(kinds[s]), is_dual[s]))
        perm = torch.tensor(order, device=dev)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(b, device=dev)
        images, text_ids, text_masks, attackable = images[perm].contiguous(), text_ids[perm], text_masks[perm], attackable[perm]
        if init_eta is not None:
            init_eta = init_eta[perm]
        kinds = [kinds[s] for s in order]
        is_dual = [is_dual[s] for s in order]
        total = [len(k) for k in kinds]
        any_dual = any(is_dual)
        if any_dual and not hasattr(a, "pgd_attack_mixed"):
            raise ValueError("dual-loss samples in attack_mixed need adapters with pgd_attack_mixed()")
        if hasattr(a, "set_mlm_samples"):          # nothing of an earlier (possibly aborted) attack may linger
            a.set_mlm_rows(None)
            a.set_mlm_samples(None)
        # ---- MLM side of the dual-loss samples: paraphrase ids / masks, labels, live label rows (fixed for the attack)
        mlm_ids = mlm_mask = labels = labels_live = emb_mlm = None
        if any_dual:
            tasks = [tasks[s] if is_dual[i] else None for i, s in enumerate(order)]
            tasks, text_ids, text_masks, attackable, mlm_ids, mlm_mask, labels = self._mixed_mlm_side(
                tasks, text_ids, text_masks, attackable)
        if proposals is None:
            fn = mlm_logits_fn or getattr(a, "mlm_logits", None)
            proposals = text_update.propose_candidates(fn(text_ids, text_masks), text_ids, attackable,
                                                       banned=self.banned_ids) if fn is not None else [[] for _ in range(b)]
        else:
            proposals = [proposals[s] for s in order]
        plan = text_update.CandidatePlan(proposals, dev)
        a.set_text(text_ids, text_masks, text_ids_mlm=mlm_ids, text_mask_mlm=mlm_mask)
        pinned = getattr(a, "_tlen", None)                                # token layout of the targets
        if any_dual:
            if getattr(a, "trim_padding", False) and pinned is not None and pinned < labels.shape[-1] and \
                    bool((labels[..., pinned:] != mlm_task.IGNORE).any()):
                raise ValueError("MLM labels beyond the batch's text length {} must be ignore_index".format(pinned))
            rows, labels_live = mlm_task.live_label_rows(labels)
            a.set_mlm_rows(rows)
            emb_mlm = ops.embed_tokens(self.tables, mlm_ids)
        targets = a.gen_ori_feats(images)
        e_ori = ops.embed_tokens(self.tables, text_ids)
        adv_emb = e_ori.clone()
        adv_ids = text_ids.clone().contiguous()
        ori_ids = text_ids.contiguous()
        rounds = []
        flag = ops.new_flag(dev)
        eta = init_eta
        if eta is None and c.random_start:
            eta = torch.empty_like(images, memory_format=torch.contiguous_format).uniform_(-c.eps, c.eps)
        cur = ops.linf_init(images.contiguous(), eta, c.eps, c.clip_min, c.clip_max, flag=flag)
        losses = torch.zeros(max(total), dtype=torch.float32, device=dev)
        ws = ops.Workspace()
        res = BatchResult(adv_images=cur, adv_text_ids=adv_ids)
        text_key, text_cache, version, sel = None, {}, 0, None            # (active samples, who is at an MLM step, text edit)
        # ALBEF re-draws a random MLM mask inside every forward that takes token ids (Gen_feats / get_mlm_logits,
        # model_pretrain.py:130-132,105-122); the probe closure takes embeddings and ignores its draw (:85-104).  Every step
        # here feeds embeddings, so the ids are masked per step and embedded for all samples that are not probing.
        masker = getattr(a, "mask_text_ids", None) if getattr(a, "random_masking", False) else None
        use_mixed_closure = hasattr(a, "pgd_attack_mixed")     # draws no mask of its own; else the plain probe closure
        ids_step = None
        for t in range(max(total)):
            n_act = sum(1 for x in total if x > t)
            now = [kinds[s][t][0] for s in range(n_act)]
            at_mlm = tuple(s for s in range(n_act) if now[s] == "M")
            key = (n_act, at_mlm, version)
            if key != text_key:
                sel = torch.tensor(at_mlm, device=dev) if at_mlm else None      # who takes an MLM step, as a device index
                if key in text_cache:
                    state, ids_step = text_cache[key]
                    a.load_text(state)
                else:
                    ids_t, masks_t = adv_ids[:n_act], text_masks[:n_act]
                    if at_mlm:
                        ids_t, masks_t = ids_t.clone(), masks_t.clone()
                        ids_t[sel], masks_t[sel] = mlm_ids[sel], mlm_mask[sel]
                    a.set_text(ids_t, masks_t, text_len=pinned)
                    ids_step = ids_t
                    if any_dual:                                          # dual iterations alternate between two texts
                        if len(text_cache) >= 2:
                            text_cache.pop(next(iter(text_cache)))
                        text_cache[key] = (a.save_text(), ids_step)
                if any_dual:
                    a.set_mlm_samples(sel)
                y = [v.rows(n_act) if isinstance(v, LayerFeatures) else (None if v is None else v[:n_act])
                     for v in self._y_feature(targets)]
                text_key = key
            leaf_img = cur[:n_act].detach().requires_grad_(True)
            probing = [s for s in range(n_act) if kinds[s][t][1]] if masker is not None else []
            emb_t = adv_emb[:n_act]
            if masker is None or probing:            # (a masked step without probing samples re-embeds every row below)
                if at_mlm:
                    emb_t = emb_t.clone()
                    emb_t[sel] = emb_mlm[sel]
            if masker is not None:
                emb_m = ops.embed_tokens(self.tables, masker(ids_step))   # one draw per step, like one per forward
                if probing:                                               # probe steps see the unmasked embeddings
                    idx = torch.tensor(probing, device=dev)
                    emb_m[idx] = emb_t[idx]
                emb_t = emb_m
            leaf_txt = emb_t.detach().requires_grad_(True)
            slot = attacks._LossSlot(losses, t)
            if use_mixed_closure:
                attacks._mixed_loss_and_grad(a.pgd_attack_mixed, [leaf_img, leaf_txt], [leaf_img, leaf_txt], list(y),
                                             self.flavor, slot, ws=ws, flag=flag,
                                             mlm_labels=labels_live[sel] if at_mlm else None)
            else:
                attacks._loss_and_grad(a.pgd_attack_vl, [leaf_img, leaf_txt], [leaf_img, leaf_txt], list(y), 1,
                                       self.flavor, False, slot, vl=True, ws=ws)
            self._update_runs(cur, attacks._grad_of(leaf_img), images, now)
            firing = [s for s in range(n_act) if kinds[s][t][1]]
            if firing:
                sub = plan.restricted_to(firing)
                scores = text_update.score_plan(self.tables, e_ori[:n_act], leaf_txt.grad, sub)
                prev = adv_ids[:n_act].clone()
                new_id, rank = text_update.accept_round(sub, scores, ori_ids[:n_act], adv_ids[:n_act],
                                                        self.similarity_fn, c.sim_threshold)   # in place, on the device
                rounds.append((prev, new_id, rank, n_act))
                ops.embed_tokens(self.tables, adv_ids, out=adv_emb)
                version += 1
                text_cache.clear()
                if any_dual and any(is_dual[s] for s in firing) and \
                        self._follow_substitutions(tasks, [s for s in firing if is_dual[s]], mlm_ids, mlm_mask,
                                                   text_update.substitution_lists(prev, new_id, rank)):
                    ops.embed_tokens(self.tables, mlm_ids, out=emb_mlm)
        if c.sanity_checks:
            bits = int(flag.item())
            assert bits == 0, "input images are outside [clip_min, clip_max]" if bits & 1 else "bad MLM label"
        # rounds in the CALLER's sample order, one entry per sample of the batch (internally the batch is sorted by schedule
        # length and a round covers the active prefix only)
        for prev, new_id, rank, n_act in rounds:
            per_round = [[] for _ in range(b)]
            for i, per in enumerate(text_update.substitution_lists(prev, new_id, rank)):
                per_round[order[i]] = per
            res.substitutions.append(per_round)
        res.adv_images, res.adv_text_ids = restore(cur[inv]), adv_ids[inv][:, :text_len]
        if any_dual:
            res.adv_text_ids_mlm = mlm_ids[inv]
        res.loss_lists = [losses.tolist()]
        res.gradient_steps = res.sample_steps = sum(total)
        res.global_steps = max(total)
        return res

    def _mixed_mlm_side(self, tasks, text_ids, text_masks, attackable):
        """MLM side of a mixed batch (``tasks``: one ``MlmTask`` per dual-loss sample, None elsewhere, in batch order).
        Returns private task copies, the text tensors widened to the longest paraphrase (ALBEF encodes a paraphrase at
        its own length, ``padding='longest'``), the paraphrase ids / masks (a feature sample's row = its question) and the
        labels (B, L) / (B, K, L) on the device (``ignore_index`` rows for feature samples and missing label sets)."""
        dev, b = text_ids.device, text_ids.shape[0]
        tasks = [None if t is None else mlm_task.MlmTask(**vars(t)) for t in tasks]
        for t in tasks:
            if t is not None:
                t.words_mlm = list(t.words_mlm)                           # private copies: update_mlm_text edits them
        width = max([text_ids.shape[1]] + [len(t.text_ids_mlm) for t in tasks if t is not None])
        if width > text_ids.shape[1]:
            grow = (0, width - text_ids.shape[1])
            text_ids, text_masks, attackable = F.pad(text_ids, grow), F.pad(text_masks, grow), F.pad(attackable, grow)
        mlm_ids, mlm_mask = text_ids.clone(), text_masks.clone()
        sets = [[] if t is None else (t.mlm_labels if isinstance(t.mlm_labels[0], list) else [t.mlm_labels])
                for t in tasks]
        k = max(len(x) for x in sets)
        labels = torch.full((b, k, width), mlm_task.IGNORE, dtype=torch.int64)
        for s, t in enumerate(tasks):
            if t is None:
                continue
            self._write_mlm_row(mlm_ids, mlm_mask, s, t)
            for j, row in enumerate(sets[s]):
                labels[s, j, :len(row)] = torch.tensor(row)
        return tasks, text_ids, text_masks, attackable, mlm_ids, mlm_mask, (labels[:, 0] if k == 1 else labels).to(dev)

    def _update_runs(self, cur, grad, images, now):
        """The fused image update of one global step, in place on the active prefix of ``cur``: one launch per run of
        consecutive samples of one kind -- ``vqa_linf_fgm`` for ``N`` (first half of a dual iteration: no projection in
        between, projected_gradient_descent.py:155-188), ``vqa_linf_step`` otherwise.  Finished samples stay untouched."""
        c, lo, n_act = self.cfg, 0, len(now)
        while lo < n_act:
            hi = lo + 1
            while hi < n_act and (now[hi] == "N") == (now[lo] == "N"):
                hi += 1
            if now[lo] == "N":
                ops.linf_fgm(cur[lo:hi], grad[lo:hi], c.eps_iter, c.clip_min, c.clip_max, out=cur[lo:hi])
            else:
                ops.linf_step(cur[lo:hi], grad[lo:hi], images[lo:hi], c.eps_iter, c.eps, c.clip_min, c.clip_max,
                              out=cur[lo:hi])
            lo = hi

    def _follow_substitutions(self, tasks, samples, mlm_ids, mlm_mask, subs):
        """``update_mlm_text`` (adv_attack.py:334-351) for the dual-loss ``samples`` that just probed: their paraphrase
        takes over the question's substitutions ``subs[s]`` = [(position, old id, new id), ...].  True if any row changed."""
        changed = False
        for s in samples:
            if subs[s]:
                mlm_task.apply_substitutions(tasks[s].words_mlm, [(old, new) for (_, old, new) in subs[s]])
                tasks[s].reencode()
                self._write_mlm_row(mlm_ids, mlm_mask, s, tasks[s])
                changed = True
        return changed

    @staticmethod
    def _write_mlm_row(mlm_ids, mlm_mask, s, task):
        """Row ``s`` of the batch's paraphrase ids / masks <- the task's current encoding (zero beyond its length)."""
        n = len(task.text_ids_mlm)
        mlm_ids[s].zero_()
        mlm_mask[s].zero_()
        mlm_ids[s, :n] = torch.tensor(task.text_ids_mlm, device=mlm_ids.device, dtype=mlm_ids.dtype)
        mlm_mask[s, :n] = torch.tensor(task.text_mask_mlm, device=mlm_mask.device, dtype=mlm_mask.dtype)
