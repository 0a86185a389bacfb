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
from . import text_update
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
    substitutions: list = field(default_factory=list)
    gradient_steps: int = 0


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
                        y=self._y_dual(targets, mlm_labels), ls=0, **common)

    @torch.no_grad()
    def attack_batch(self, images, text_ids, text_masks, attackable, mlm_logits_fn=None, dual=False,
                     mlm_labels=None, init_eta=None, proposals=None, text_ids_mlm=None):
        """Attack one batch whose samples all have the same number of attackable words.

        images (B,3,H,W) in [clip_min, clip_max]; text_ids/text_masks (B,L); attackable bool (B,L) with the same
        count per row.  ``mlm_logits_fn(text_ids, text_masks) -> (B,L,V)`` proposes substitution candidates (the
        reference uses a separate HF BERT-MLM, adv_attack.py:110,242; default: the adapters' own ``mlm_logits``);
        ``proposals`` injects them directly (``text_update.propose_candidates`` format).  Returns a ``BatchResult``.

        ``dual=True`` runs the reference's ``old_alg == 0`` blocks (feature step + MLM step per iteration,
        adv_attack.py:614-619,670-676) with ``mlm_labels`` (B, L) / (B, K, L); ``text_ids_mlm`` is the [MASK]-ed
        paraphrase the MLM closure reads (``self.batch["text_ids_mlm"]``), kept position-aligned with the question:
        accepted substitutions are applied to it too, like ``update_mlm_text`` (:334-351).
        """
        c, a = self.cfg, self.adapters
        n_words = int(attackable[0].sum().item())
        if not bool((attackable.sum(dim=1) == n_words).all()):
            raise ValueError("samples of one batch must share a schedule: bucket them with bucket_by_schedule()")
        blocks = iter_schedule(n_words, c.budget)
        images, init_eta, restore = self._enter_layout(images, init_eta)
        a.set_text(text_ids, text_masks)
        targets = a.gen_ori_feats(images)
        adv = images
        adv_ids = text_ids.clone()
        mlm_ids = None if text_ids_mlm is None else text_ids_mlm.clone()
        if mlm_ids is not None:
            a.set_text(text_ids, text_masks, text_ids_mlm=mlm_ids)
        res = BatchResult(adv_images=adv, adv_text_ids=adv_ids)
        first_time = 0 if c.random_start else 1
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
            # text embeddings of the question batch: one launch; after every substitution round only the rows of the
            # replaced words are rewritten (masked-token embedding substitution, ops.embed_tokens)
            e_ori = ops.embed_tokens(self.tables, text_ids)
            adv_emb = e_ori.clone()
            ori_host = text_ids.cpu().numpy()
            positions = list(range(text_ids.shape[1]))
            for bi, steps in enumerate(blocks):
                a.set_text(adv_ids, text_masks, text_ids_mlm=mlm_ids)
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
                if proposals is not None:
                    cand, scores = text_update.score_candidates(self.tables, e_ori, text_grad, proposals)
                    new_ids, subs = text_update.greedy_accept(cand, scores, ori_host, adv_ids.cpu().numpy(),
                                                              self.similarity_fn, c.sim_threshold)
                    adv_ids = torch.as_tensor(new_ids, device=text_ids.device, dtype=text_ids.dtype)
                    res.substitutions.append(subs)
                    changed = [(s, p) for s, per in enumerate(subs) for (p, _, _) in per]
                    if changed:
                        ops.embed_tokens(self.tables, adv_ids, out=adv_emb, rows=changed)
                        if mlm_ids is not None:      # update_mlm_text: same word replaced in the MLM paraphrase
                            for s, per in enumerate(subs):
                                for (p, old, new) in per:
                                    if int(mlm_ids[s, p]) == old:
                                        mlm_ids[s, p] = new
        res.adv_images, res.adv_text_ids = restore(adv), adv_ids
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
        a.set_text(text_ids, text_masks)
        pinned = getattr(a, "_tlen", None)                                # token layout of the targets
        targets = a.gen_ori_feats(images)
        e_ori = ops.embed_tokens(self.tables, text_ids)
        adv_emb = e_ori.clone()
        adv_ids = text_ids.clone()
        ori_host = text_ids.cpu().numpy()
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
                props = [proposals[s] if s in firing else [] for s in range(n_act)]
                cand, scores = text_update.score_candidates(self.tables, e_ori[:n_act], leaf_txt.grad, props)
                new_ids, subs = text_update.greedy_accept(cand, scores, ori_host[:n_act], adv_ids[:n_act].cpu().numpy(),
                                                          self.similarity_fn, c.sim_threshold)
                changed = [(s, p) for s, per in enumerate(subs) for (p, _, _) in per]
                if changed:
                    adv_ids[:n_act] = torch.as_tensor(new_ids, device=dev, dtype=adv_ids.dtype)
                    ops.embed_tokens(self.tables, adv_ids, out=adv_emb, rows=changed)
                    a.set_text(adv_ids[:n_act], text_masks[:n_act], text_len=pinned)
                res.substitutions.append(subs)
        if c.sanity_checks:
            assert int(flag.item()) == 0, "input images are outside [clip_min, clip_max]"
        res.adv_images, res.adv_text_ids = restore(cur[inv]), adv_ids[inv]
        res.loss_lists = [losses.tolist()]
        res.gradient_steps = sum(total)
        return res
