"""The reference's OWN ``model_fn`` closures over the bundled frozen white boxes -- what a user of the reference has.

The batched drivers of this package (``attack/runner.py``) hand the operators ``LayerFeatures`` (per-layer maps left
where the encoder wrote them, batch > 1, row weights).  A user who switches only the ``cleverhans`` import
 keeps the reference's orchestrator instead, whose closures are members of the attack class
that read ``self.batch``, run ONE sample and hand back ``torch.stack`` / ``torch.cat``-packed plain tensors:

  * VLMO  ``pgd_attack`` vlmo_module.py:1387-1446 -> ``[pooler(cls) (1, D), cls per layer (depth+1, D),
          tokens per layer (depth+1, n_real_text + n_image, D)]`` (``[0]`` indexing: batch 1; padded text tokens dropped
          with ``torch.where``; the image part starts at the literal text length 40), ``pgd_attack_vl`` :1328-1385
          (``x = [image, text_embeds]``), ``pgd_mlm_attack`` :1448-1529 (``[mlm logits, cls per layer, tokens]``),
          ``Gen_ori_feats`` :1287-1312;
  * ALBEF ``pgd_attack`` adv_attack.py:119-126 -> ``[cat(text maps, 0) (13 B, L, D), cat(image maps, 0) (13 B, N, D)]``,
          ``pgd_attack_vl`` :208-214, ``pgd_mlm_attack`` :130-140 (``[mlm logits]``), ``Gen_ori_feats`` :111-118.

These classes are that form, written against ``FrozenVlmo`` / ``FrozenAlbef``: a mutable ``batch`` dict the caller
updates between operator calls (``adv_attack.py:631-632``), plain-tensor lists out.  They exist so that the drop-in is
exercised and timed the way the reference would drive it (``tests/test_reference_style_dropin.py``,
``bench.py --reference-style``); the packing copies they make are the reference's, not something the batched path does.
"""
import torch


class VlmoReferenceClosures:
    """``self.batch``: ``text_ids`` / ``text_masks`` (1, L) and, for the MLM closure, ``text_ids_mlm`` /
    ``text_mask_mlm`` -- the keys the reference's closures read.

    What depends on the text batch only is taken ONCE per text batch (keyed on the mask tensor's address and in-place
    version), not once per forward as in the reference: the index of the real text tokens (``torch.where``: a
    device->host read, vlmo_module.py:1441) and the per-layer attention masks (``get_rel_pos_bias`` + key padding,
    :1416,88-118).  With a prefix mask (``[CLS] words [SEP] pad...``: what a tokenizer produces) the padded columns are
    not run through the encoder at all -- they are masked keys and dropped rows, so no result changes -- and the packed
    per-token tensor is the stacked maps themselves (no gather, no ``cat``).  The closures then contain no host read and
    can be captured into a hipGraph: ``pgd(..., graph=True)``; ``capturable`` says so."""

    capturable = True

    def __init__(self, model, batch):
        self.model, self.batch = model, batch
        self._text = None

    def _text_state(self, text_masks):
        key = (text_masks.data_ptr(), text_masks._version, tuple(text_masks.shape))
        if self._text is None or self._text[0] != key:
            m = text_masks[0].bool()
            n = int(m.sum())                                                   # the one host read per text batch
            prefix = bool(m[:n].all())
            real = None if prefix else torch.where(text_masks[0] == 1)[0]
            keep = n if prefix else text_masks.shape[1]
            bias = self.model.attention_bias(text_masks[:, :keep])
            self._text = (key, keep, real, bias)
        return self._text[1:]

    def _encode(self, image, text_embeds, text_masks):
        keep, real, bias = self._text_state(text_masks)
        feats, states = self.model.encode(image, text_embeds[:, :keep], text_masks[:, :keep], bias)
        target = torch.stack(feats, axis=1)                                  # (1, depth + 1, keep + N, D)
        if real is None:                                                     # prefix mask: rows = real text + image
            per_token = target[0]
        else:
            per_token = torch.cat([target[0, :, real], target[0, :, self.model.cfg.max_text_len:]], axis=1)
        return states, self.model.pooled(states), target[0, :, 0, :], per_token, keep

    def pgd_attack(self, x):
        ids, masks = self.batch["text_ids"], self.batch["text_masks"]
        _, pooled, per_layer_cls, per_token, _ = self._encode(x, self.model.text_embeddings(ids), masks)
        return [pooled, per_layer_cls, per_token]

    def pgd_attack_vl(self, x):
        _, pooled, per_layer_cls, per_token, _ = self._encode(x[0], x[1], self.batch["text_masks"])
        return [pooled, per_layer_cls, per_token]

    def pgd_mlm_attack(self, x):
        ids, masks = self.batch["text_ids_mlm"], self.batch["text_mask_mlm"]
        states, _, per_layer_cls, per_token, keep = self._encode(x, self.model.text_embeddings(ids), masks)
        logits = self.model.mlm_score(states[:, :keep])
        full = self.model.cfg.max_text_len
        if keep < full:             # the reference's (1, 40, V) contract: labels at the dropped positions are ignore_index
            logits = torch.nn.functional.pad(logits, (0, 0, 0, full - keep))
        return [logits, per_layer_cls, per_token]

    def Gen_ori_feats(self, image):
        with torch.no_grad():
            return [t.detach() for t in self.pgd_attack(image)]


class AlbefReferenceClosures:
    """ALBEF's closures (adv_attack.py:111-126,130-140,208-214).  Capturable as well: the per-forward random token masking
    (model_pretrain.py:130-132) is drawn on the device from torch's graph-safe generator when no masking seed is set."""

    capturable = True

    def __init__(self, model, batch):
        self.model, self.batch = model, batch

    def pgd_attack(self, x):
        img, txt = self.model.gen_feats(x, self.batch["text_ids"], self.batch["text_masks"])
        return [torch.cat(txt, axis=0), torch.cat(img, axis=0)]

    def pgd_attack_vl(self, x):
        img, txt = self.model.gen_feats_from_embeds(x[0], x[1], self.batch["text_ids"], self.batch["text_masks"])
        return [torch.cat(txt, axis=0), torch.cat(img, axis=0)]

    def pgd_mlm_attack(self, x):
        return [self.model.get_mlm_logits(x, self.batch["text_ids_mlm"], self.batch["text_mask_mlm"])]

    def Gen_ori_feats(self, image):
        with torch.no_grad():
            txt, img = self.pgd_attack(image)
        return img.detach(), txt.detach()                                    # the reference's order: image first
