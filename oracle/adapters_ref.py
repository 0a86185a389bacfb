"""TEST INFRASTRUCTURE ONLY -- reference-style (batch-1, packed) white-box adapters for the CPU oracle.

The shipped adapters (``vqattack_amd/whitebox``) are batched and never pack features; the reference's are batch-1 and
pack with ``torch.stack`` / ``torch.cat``.  These closures restate the reference's packing over the SAME frozen encoder
object so that (a) the oracle's loss sees exactly the tensors the reference's FGM would see and (b) bench.py's
``cpu_baseline`` times the reference's own structure.  Restated from source text (the orchestrators cannot be imported
here), hence **parity unpinned** for the packing itself; the arithmetic downstream is the pinned oracle.

  * VLMO:  ``pgd_attack`` vlmo_module.py:1387-1446, ``pgd_attack_vl`` :1328-1385, ``pgd_mlm_attack`` :1448-1529,
           ``Gen_ori_feats`` :1287-1312
  * ALBEF: ``pgd_attack`` adv_attack.py:119-126, ``pgd_attack_vl`` :208-214, ``pgd_mlm_attack`` :130-140,
           ``Gen_ori_feats`` :111-118
"""
import torch


class VlmoRefAdapters:
    """Batch 1 only (the reference indexes ``[0]``)."""

    def __init__(self, model, text_ids, text_masks):
        assert text_ids.shape[0] == 1
        self.m, self.ids, self.masks = model, text_ids, text_masks

    def _pack(self, feats, states):
        stacked = torch.stack(feats, axis=1)                         # (1, depth+1, T+N, D)
        tlen = self.m.cfg.max_text_len
        img = stacked[0, :, tlen:]
        keep = torch.where(self.masks[0] == 1)[0]
        txt = stacked[0, :, keep]
        return [self.m.pooled(states), stacked[0, :, 0, :], torch.cat([txt, img], axis=1)]

    def pgd_attack(self, x):
        feats, states = self.m.encode(x, self.m.text_embeddings(self.ids), self.masks)
        return self._pack(feats, states)

    def pgd_attack_vl(self, xs):
        feats, states = self.m.encode(xs[0], xs[1], self.masks)
        return self._pack(feats, states)

    def pgd_mlm_attack(self, x):
        feats, states = self.m.encode(x, self.m.text_embeddings(self.ids), self.masks)
        packed = self._pack(feats, states)
        return [self.m.mlm_score(states[:, :self.m.cfg.max_text_len]), packed[1], packed[2]]

    def gen_ori_feats(self, image):
        with torch.no_grad():
            out = self.pgd_attack(image)
        return [out[0].detach(), out[1].detach(), out[2].detach()]


class AlbefRefAdapters:
    def __init__(self, model, text_ids, text_masks):
        self.m, self.ids, self.masks = model, text_ids, text_masks

    def pgd_attack(self, x):
        img, txt = self.m.gen_feats(x, self.ids, self.masks)
        return [torch.cat(txt, axis=0), torch.cat(img, axis=0)]

    def pgd_attack_vl(self, xs):
        img, txt = self.m.gen_feats_from_embeds(xs[0], xs[1], self.ids, self.masks)
        return [torch.cat(txt, axis=0), torch.cat(img, axis=0)]

    def pgd_mlm_attack(self, x):
        return [self.m.get_mlm_logits(x, self.ids, self.masks)]

    def gen_ori_feats(self, image):
        with torch.no_grad():
            out = self.pgd_attack(image)
        return [out[0].detach(), out[1].detach()]
