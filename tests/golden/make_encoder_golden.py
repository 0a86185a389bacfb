"""White-box encoder fixtures from the REFERENCE's own model classes (build container only).

What runs here is the reference's code, compiled from its source files by ``refexec`` (nothing rewritten):

  * VLMo: ``Mlp`` / ``Attention`` / ``Block`` / ``PatchEmbed`` / ``MultiWayTransformer`` of
    ``VLMO_VQAttack/vlmo/modules/multiway_transformer.py:33-383``, ``Pooler`` / ``MLMHead`` of ``modules/heads.py``,
    and the ``VLMo`` methods ``build_relative_position_embed`` / ``get_rel_pos_bias`` (vlmo_module.py:806-880) and the three
    attack closures ``pgd_attack`` / ``pgd_attack_vl`` / ``pgd_mlm_attack`` (:1328-1529) bound to a module that holds
    exactly those parts under the reference's attribute names -- so ``state_dict()`` has the reference's keys;
  * ALBEF: ``Mlp`` / ``Attention`` / ``Block`` / ``VisionTransformer`` / ``interpolate_pos_embed`` of
    ``ALBEF_attack/models/vit.py`` and ``BertEmbeddings`` ... ``BertEncoder`` + the MLM head classes of
    ``models/xbert.py:169-700``; the glue between them is ``BertModel.forward`` (xbert.py:1022-1075: additive masks
    ``(1 - mask) * -10000``) and ``BertForMaskedLM.forward`` (:1417-1466), restated in ``_albef_forward`` below because
    those two classes derive from ``transformers.PreTrainedModel`` of the pinned 4.8.1, not of this image's 5.x.

Third-party names the reference imports and this image lacks (timm 0.4.12, ``V/requirements.txt:10``) are bound to the
equivalent objects of installed libraries, never to code written here, with one exception:
``to_2tuple`` -> ``torch.nn.modules.utils._pair``; ``trunc_normal_`` -> ``torch.nn.init.trunc_normal_`` (initialisation
only; every weight is overwritten by the seeded filler); ``DropPath`` is never constructed (``drop_path == 0`` takes the
``nn.Identity`` branch, multiway_transformer.py:147); timm's ``PatchEmbed`` (ALBEF's ViT only) is the 4-line module
``_TimmPatchEmbed`` below: ``Conv2d(kernel = stride = patch)`` then ``flatten(2).transpose(1, 2)`` (timm 0.4.12
``vision_transformer.py``).  VLMo's text embeddings are ``transformers==4.8.1``'s ``BertEmbeddings`` (``V/requirements.txt:1``),
whose source the reference vendors at ``ALBEF_attack/models/xbert.py:169-216`` -- that copy is executed (this image's
transformers 5.x dropped ``position_embedding_type``, which VLMo sets to "rel_pos", vlmo_module.py:216).

Outputs are data only (``encoder_golden.npz`` / ``.json``): inputs are re-derived from seeds (``encoder_cases.py``), the
reference-format state dict is a (key, shape) listing + seed + checksums (verbatim weights for the small cases), the
expected values are the reference's outputs and input gradients (sub-sampled rows / elements at base width).
"""
import functools
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import encoder_cases as ec          # noqa: E402
from tests.golden import refexec as rx                # noqa: E402

MWT = rx.REF + "/VLMO_VQAttack/vlmo/modules/multiway_transformer.py"
VLMO_HEADS = rx.REF + "/VLMO_VQAttack/vlmo/modules/heads.py"
ALBEF_VIT = rx.REF + "/ALBEF_VQAttack/ALBEF_attack/models/vit.py"


def _hf_globals():
    from transformers.activations import ACT2FN
    from transformers.modeling_outputs import BaseModelOutputWithPastAndCrossAttentions
    from transformers.pytorch_utils import apply_chunking_to_forward, prune_linear_layer
    return dict(ACT2FN=ACT2FN, BaseModelOutputWithPastAndCrossAttentions=BaseModelOutputWithPastAndCrossAttentions,
                apply_chunking_to_forward=apply_chunking_to_forward, prune_linear_layer=prune_linear_layer,
                logger=rx.namespace(warn=print, info=lambda *a, **k: None))


XBERT_CLASSES = ["BertEmbeddings", "BertSelfAttention", "BertSelfOutput", "BertAttention", "BertIntermediate", "BertOutput",
                 "BertLayer", "BertEncoder", "BertPredictionHeadTransform", "BertLMPredictionHead", "BertOnlyMLMHead"]


def _bert_config(**kw):
    base = dict(pad_token_id=0, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.0,
                attention_probs_dropout_prob=0.0, hidden_act="gelu", chunk_size_feed_forward=0, add_cross_attention=False)
    base.update(kw)
    return rx.namespace(**base)


# ---- VLMo ----------------------------------------------------------------------------------------------------------
def build_ref_vlmo(case):
    mwt = rx.module_items(MWT, ["Mlp", "Attention", "Block", "PatchEmbed", "MultiWayTransformer"],
                          extra_globals=dict(partial=functools.partial, to_2tuple=nn.modules.utils._pair,
                                             trunc_normal_=nn.init.trunc_normal_, rank_zero_info=lambda *a, **k: None))
    xb = rx.module_items(rx.ALBEF_XBERT, ["BertEmbeddings", "BertPredictionHeadTransform"], extra_globals=_hf_globals())
    heads = rx.module_items(VLMO_HEADS, ["Pooler", "MLMHead"],
                            extra_globals=dict(BertPredictionHeadTransform=xb["BertPredictionHeadTransform"]))
    methods, _ = rx.class_methods(rx.VLMO_MODULE, "VLMo", ["build_relative_position_embed", "get_rel_pos_bias",
                                                           "pgd_attack", "pgd_attack_vl", "pgd_mlm_attack"])
    ref_cls = type("RefVLMo", (nn.Module,), dict(methods))
    d = case["dim"]
    config = dict(drop_path_rate=0.0, loss_names=dict(textmlm=0), max_text_len=case["max_text_len"],
                  max_text_len_of_initckpt=196)
    m = ref_cls()
    m.img_size = case["image_size"]
    m.transformer = mwt["MultiWayTransformer"](
        img_size=case["image_size"], patch_size=case["patch"], embed_dim=d, depth=case["depth"], num_heads=case["heads"],
        mlp_ratio=4, qkv_bias=True, vlffn_start_layer_index=case["vlffn_start"],
        norm_layer=functools.partial(nn.LayerNorm, eps=1e-6), config=config)
    m.patch_size = m.transformer.patch_size
    m.num_layers = len(m.transformer.blocks)
    m.num_features = d
    m.build_relative_position_embed(config)
    bert_config = _bert_config(vocab_size=case["vocab"], hidden_size=d, max_position_embeddings=case["max_text_len"],
                               position_embedding_type="rel_pos")                       # vlmo_module.py:210-217
    m.text_embeddings = xb["BertEmbeddings"](bert_config)
    m.token_type_embeddings = nn.Embedding(2, d)
    m.pooler = heads["Pooler"](d)
    m.mlm_score = heads["MLMHead"](bert_config)
    m.vqa_classifier = nn.Sequential(nn.Linear(d, d * 2), nn.LayerNorm(d * 2), nn.GELU(),
                                     nn.Linear(d * 2, case["n_answers"]))               # vlmo_module.py:274-279
    m.batch = {}
    return m.eval()


def _scalar(outs, weights):
    return sum((o * w).sum() for o, w in zip(outs, weights))


def vlmo_case(name, case, arrays, meta):
    torch.manual_seed(0)
    ref = build_ref_vlmo(case)
    listing = ec.listing_of(ref.state_dict())
    sd = ec.seeded_state_dict(listing, case["seed"])
    ref.load_state_dict(sd, strict=True)
    for p in ref.parameters():
        p.requires_grad_(False)
    inp = ec.case_inputs(name, case, "vlmo")
    rec = dict(flavor="vlmo", listing=listing, checksums=ec.checksums(sd), seed=case["seed"], samples=[])
    if case["store_weights"]:
        for k, v in sd.items():
            arrays["{}/sd/{}".format(name, k)] = v.numpy()
    small = case["store_weights"]
    with rx.cpu_as_cuda():
        for b in range(2):
            image = inp["image"][b:b + 1]
            ref.batch = {"text_ids": inp["ids"][b:b + 1], "text_masks": inp["masks"][b:b + 1],
                         "text_ids_mlm": inp["mlm_ids"][b:b + 1], "text_mask_mlm": inp["masks"][b:b + 1]}
            x = image.clone().requires_grad_(True)
            outs = ref.pgd_attack(x)
            w = ec.functional_weights([o.shape for o in outs], case["seed"] + b)
            _scalar(outs, w).backward()
            g_img = x.grad.detach().clone()
            # image + text-embedding closure
            emb = ref.text_embeddings(ref.batch["text_ids"]).detach()
            x2, e2 = image.clone().requires_grad_(True), emb.clone().requires_grad_(True)
            outs_vl = ref.pgd_attack_vl([x2, e2])
            _scalar(outs_vl, w).backward()
            # MLM closure
            x3 = image.clone().requires_grad_(True)
            outs_mlm = ref.pgd_mlm_attack(x3)
            n_real = case["text_lens"][b]                             # padded positions carry no label in the attack
            w_mlm = ec.functional_weights([outs_mlm[0][:, :n_real].shape], case["seed"] + 50 + b)
            _scalar([outs_mlm[0][:, :n_real]], w_mlm).backward()
            with torch.no_grad():
                cls_feats = outs[0].detach()
                vqa_logits = ref.vqa_classifier(cls_feats)
            key = "{}/s{}/".format(name, b)
            feats = outs[2].detach()                                   # (depth + 1, n_real + N, D)
            rows = ec.sample_rows(feats.shape[1], case["seed"] + b) if not small else torch.arange(feats.shape[1])
            flat = ec.sample_flat(g_img.numel(), case["seed"] + b) if not small else torch.arange(g_img.numel())
            arrays[key + "cls_feats"] = cls_feats.numpy()
            arrays[key + "cls_per_layer"] = outs[1].detach().numpy()
            arrays[key + "feats_rows"] = rows.numpy()
            arrays[key + "feats"] = feats[:, rows].numpy()
            arrays[key + "feats_norm"] = feats.double().pow(2).sum(dim=(1, 2)).sqrt().numpy()
            arrays[key + "grad_index"] = flat.numpy().astype(np.int32)
            arrays[key + "grad_image"] = g_img.reshape(-1)[flat].numpy()
            arrays[key + "grad_image_norm"] = np.asarray(float(g_img.double().norm()))
            arrays[key + "vl_grad_image"] = x2.grad.reshape(-1)[flat].numpy()
            arrays[key + "vl_grad_text"] = e2.grad[0].numpy()           # (40, D): zero rows at the padded positions
            arrays[key + "text_embeds"] = emb[0].numpy()
            arrays[key + "mlm_logits"] = outs_mlm[0].detach()[0, :case["text_lens"][b]].numpy()
            arrays[key + "mlm_grad_image"] = x3.grad.reshape(-1)[flat].numpy()
            arrays[key + "vqa_logits"] = vqa_logits.numpy()
            rec["samples"].append(dict(n_text=int(case["text_lens"][b]), n_rows=int(feats.shape[1]),
                                       loss=float(_scalar(outs, w).detach())))
    # the additive bias of every layer, as the reference's attention receives it (get_rel_pos_bias)
    with torch.no_grad():
        bias = torch.stack(list(ref.get_rel_pos_bias(ref.text_imag_relative_position_index)))   # (depth, H, S, S)
    s = bias.shape[-1]
    pick = ec.sample_rows(s, case["seed"] + 77, k=4)
    arrays[name + "/rel_pos_rows"] = pick.numpy()
    arrays[name + "/rel_pos_bias_rows"] = bias[:, :, pick].numpy() if not small else bias.numpy()
    arrays[name + "/rel_pos_index"] = ref.text_imag_relative_position_index.long().numpy().astype(np.int32) \
        if small else ref.text_imag_relative_position_index.long()[pick].numpy().astype(np.int32)
    meta[name] = rec


# ---- ALBEF ---------------------------------------------------------------------------------------------------------
class _TimmPatchEmbed(nn.Module):
    """timm 0.4.12 ``vision_transformer.PatchEmbed`` (the ALBEF ViT's only timm module): the one stand-in of this file."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def build_ref_albef(case):
    vit = rx.module_items(ALBEF_VIT, ["Mlp", "Attention", "Block", "VisionTransformer", "interpolate_pos_embed"],
                          extra_globals=dict(partial=functools.partial, PatchEmbed=_TimmPatchEmbed,
                                             trunc_normal_=nn.init.trunc_normal_))
    xb = rx.module_items(rx.ALBEF_XBERT, XBERT_CLASSES, extra_globals=_hf_globals())
    d = case["dim"]
    cfg = _bert_config(vocab_size=case["vocab"], hidden_size=d, max_position_embeddings=64, num_attention_heads=case["heads"],
                       intermediate_size=4 * d, num_hidden_layers=case["bert_depth"], fusion_layer=case["fusion_layer"],
                       encoder_width=d, position_embedding_type="absolute")       # configs/config_bert.json
    m = nn.Module()
    m.visual_encoder = vit["VisionTransformer"](img_size=case["image_size"], patch_size=case["patch"], embed_dim=d,
                                                depth=case["vit_depth"], num_heads=case["heads"], mlp_ratio=4, qkv_bias=True,
                                                norm_layer=functools.partial(nn.LayerNorm, eps=1e-6))
    m.text_encoder = nn.Module()
    m.text_encoder.bert = nn.Module()
    m.text_encoder.bert.embeddings = xb["BertEmbeddings"](cfg)
    m.text_encoder.bert.encoder = xb["BertEncoder"](cfg)
    m.text_encoder.cls = xb["BertOnlyMLMHead"](cfg)
    m.text_encoder.cls.predictions.decoder.weight = m.text_encoder.bert.embeddings.word_embeddings.weight   # tie_weights
    return m.eval(), vit["interpolate_pos_embed"]


def _albef_forward(m, image, ids, masks, text_embeds=None):
    """``Gen_feats`` / ``Gen_feats_from_embeds`` / ``get_mlm_logits`` (model_pretrain.py:85-141) with the random token
    masking switched off (``mlm_probability`` 0: the draw is pinned separately, tests/test_albef_masking.py)."""
    image_embeds, img_feats = m.visual_encoder(image)
    bert = m.text_encoder.bert
    emb = bert.embeddings(input_ids=ids) if text_embeds is None else text_embeds        # xbert.py:1046-1055
    ext = (1.0 - masks[:, None, None, :].to(emb.dtype)) * -10000.0                     # get_extended_attention_mask :929-943
    img_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long)
    enc_ext = (1.0 - img_atts[:, None, None, :].to(emb.dtype)) * -10000.0              # invert_attention_mask
    out, txt_feats = bert.encoder(emb, attention_mask=ext, encoder_hidden_states=image_embeds,
                                  encoder_attention_mask=enc_ext, return_dict=True, mode="multi_modal")
    logits = m.text_encoder.cls(out.last_hidden_state)                                 # BertForMaskedLM.forward :1448-1449
    return img_feats, txt_feats, logits


def albef_case(name, case, arrays, meta):
    torch.manual_seed(0)
    ref, interp = build_ref_albef(case)
    full = ref.state_dict()
    listing = [e for e in ec.listing_of(full) if not e[0].endswith("cls.predictions.decoder.weight")]
    sd = ec.seeded_state_dict(listing, case["seed"])
    sd["text_encoder.cls.predictions.decoder.weight"] = sd["text_encoder.bert.embeddings.word_embeddings.weight"]
    sd["text_encoder.cls.predictions.decoder.bias"] = sd["text_encoder.cls.predictions.bias"]
    ref.load_state_dict(sd, strict=True)
    for p in ref.parameters():
        p.requires_grad_(False)
    inp = ec.case_inputs(name, case, "albef")
    listing = ec.listing_of(sd)
    rec = dict(flavor="albef", listing=listing, checksums=ec.checksums(sd), seed=case["seed"], samples=[],
               tied=["text_encoder.cls.predictions.decoder.weight", "text_encoder.cls.predictions.decoder.bias"])
    small = case["store_weights"]
    if small:
        for k, v in sd.items():
            if k not in rec["tied"]:
                arrays["{}/sd/{}".format(name, k)] = v.numpy()
    for b in range(2):
        image = inp["image"][b:b + 1]
        ids, masks = inp["ids"][b:b + 1], inp["masks"][b:b + 1]
        x = image.clone().requires_grad_(True)
        img_feats, txt_feats, logits = _albef_forward(ref, x, ids, masks)
        img, txt = torch.cat(img_feats, 0), torch.cat(txt_feats, 0)      # adv_attack.py:124-125: (L + 1, N, D)
        n = case["text_lens"][b]
        w = ec.functional_weights([txt[:, :n].shape, img.shape], case["seed"] + b)
        _scalar([txt[:, :n], img], w).backward()
        g_img = x.grad.detach().clone()
        x2 = image.clone().requires_grad_(True)
        emb = ref.text_encoder.bert.embeddings(input_ids=ids).detach()
        e2 = emb.clone().requires_grad_(True)
        img2, txt2, _ = _albef_forward(ref, x2, ids, masks, text_embeds=e2)
        _scalar([torch.cat(txt2, 0)[:, :n], torch.cat(img2, 0)], w).backward()
        x3 = image.clone().requires_grad_(True)
        _, _, logits3 = _albef_forward(ref, x3, inp["mlm_ids"][b:b + 1], masks)
        w_mlm = ec.functional_weights([logits3[:, :n].shape], case["seed"] + 50 + b)
        _scalar([logits3[:, :n]], w_mlm).backward()
        key = "{}/s{}/".format(name, b)
        rows = ec.sample_rows(img.shape[1], case["seed"] + b) if not small else torch.arange(img.shape[1])
        flat = ec.sample_flat(g_img.numel(), case["seed"] + b) if not small else torch.arange(g_img.numel())
        arrays[key + "img_rows"] = rows.numpy()
        arrays[key + "img_feats"] = img.detach()[:, rows].numpy()
        arrays[key + "img_feats_norm"] = img.detach().double().pow(2).sum(dim=(1, 2)).sqrt().numpy()
        arrays[key + "txt_feats"] = txt.detach()[:, :n].numpy()
        arrays[key + "grad_index"] = flat.numpy().astype(np.int32)
        arrays[key + "grad_image"] = g_img.reshape(-1)[flat].numpy()
        arrays[key + "grad_image_norm"] = np.asarray(float(g_img.double().norm()))
        arrays[key + "vl_grad_image"] = x2.grad.reshape(-1)[flat].numpy()
        arrays[key + "vl_grad_text"] = e2.grad[0].numpy()
        arrays[key + "text_embeds"] = emb[0].numpy()
        arrays[key + "mlm_logits"] = logits3.detach()[0, :n].numpy()
        arrays[key + "mlm_grad_image"] = x3.grad.reshape(-1)[flat].numpy()
        rec["samples"].append(dict(n_text=int(n), n_rows=int(img.shape[1])))
    # the position-table resampling (vit.py:181-205): the case's own table taken to a 1.5x grid by the reference's function
    grid = case["image_size"] // case["patch"]
    dst = grid * 3 // 2
    stub = rx.namespace(patch_embed=rx.namespace(num_patches=dst * dst), pos_embed=torch.zeros(1, dst * dst + 1, case["dim"]))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        resampled = interp(sd["visual_encoder.pos_embed"], stub)
    keep = ec.sample_rows(resampled.shape[1], case["seed"] + 33, k=32)
    arrays[name + "/pos_embed_dst_tokens"] = np.asarray(dst * dst + 1)
    arrays[name + "/pos_embed_rows"] = keep.numpy()
    arrays[name + "/pos_embed_resampled"] = resampled[0, keep].numpy()
    meta[name] = rec


# ---- ALBEF VQA victim: fusion encoder + answer decoder + rank_answer ------------------------------------------------
def build_ref_albef_vqa(case):
    """``ALBEF`` of models/model_vqa.py:12-45 from the reference's classes: ``visual_encoder`` (VisionTransformer),
    ``text_encoder`` (BertModel: embeddings + encoder, fusion from ``fusion_layer``), ``text_decoder`` (BertLMHeadModel:
    ``bert.embeddings`` + ``bert.encoder`` with ``fusion_layer = 0`` and ``num_hidden_layers = 6`` there, model_vqa.py:30-33,
    + ``cls``), under the attribute names that give the fine-tuned checkpoint's state-dict keys."""
    vit = rx.module_items(ALBEF_VIT, ["Mlp", "Attention", "Block", "VisionTransformer"],
                          extra_globals=dict(partial=functools.partial, PatchEmbed=_TimmPatchEmbed,
                                             trunc_normal_=nn.init.trunc_normal_))
    xb = rx.module_items(rx.ALBEF_XBERT, XBERT_CLASSES, extra_globals=_hf_globals())
    d = case["dim"]
    common = dict(vocab_size=case["vocab"], hidden_size=d, max_position_embeddings=64, num_attention_heads=case["heads"],
                  intermediate_size=4 * d, encoder_width=d, position_embedding_type="absolute")
    enc_cfg = _bert_config(num_hidden_layers=case["bert_depth"], fusion_layer=case["fusion_layer"], **common)
    dec_cfg = _bert_config(num_hidden_layers=case["dec_depth"], fusion_layer=0, **common)
    m = nn.Module()
    m.visual_encoder = vit["VisionTransformer"](img_size=case["image_size"], patch_size=case["patch"], embed_dim=d,
                                                depth=case["vit_depth"], num_heads=case["heads"], mlp_ratio=4, qkv_bias=True,
                                                norm_layer=functools.partial(nn.LayerNorm, eps=1e-6))
    m.text_encoder = nn.Module()
    m.text_encoder.embeddings = xb["BertEmbeddings"](enc_cfg)
    m.text_encoder.encoder = xb["BertEncoder"](enc_cfg)
    m.text_decoder = nn.Module()
    m.text_decoder.bert = nn.Module()
    m.text_decoder.bert.embeddings = xb["BertEmbeddings"](dec_cfg)
    m.text_decoder.bert.encoder = xb["BertEncoder"](dec_cfg)
    m.text_decoder.cls = xb["BertOnlyMLMHead"](dec_cfg)
    m.text_decoder.cls.predictions.decoder.weight = m.text_decoder.bert.embeddings.word_embeddings.weight
    return m.eval()


def _ext(mask, dtype):
    return (1.0 - mask[:, None, None, :].to(dtype)) * -10000.0


def _ref_question_states(m, image, ids, masks):
    """``ALBEF.forward(train=False)`` up to the question states (model_vqa.py:122-128; BertModel.forward glue)."""
    image_embeds, _ = m.visual_encoder(image)
    emb = m.text_encoder.embeddings(input_ids=ids)
    img_atts = torch.ones(image_embeds.shape[:-1], dtype=torch.long)
    out, _ = m.text_encoder.encoder(emb, attention_mask=_ext(masks, emb.dtype), encoder_hidden_states=image_embeds,
                                    encoder_attention_mask=_ext(img_atts, emb.dtype), return_dict=True, mode="multi_modal")
    return out.last_hidden_state


def _ref_text_decoder(m):
    """``BertLMHeadModel.forward`` (xbert.py:1219-1330) around the reference's decoder classes: causal x padding mask
    (``get_extended_attention_mask`` :905-943 with ``is_decoder``), inverted encoder mask, shifted per-token CE summed over
    the sequence (``reduction='none'``)."""
    import torch.nn.functional as F

    def text_decoder(input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                     return_dict=True, reduction="none"):
        n, length = input_ids.shape
        atts = torch.ones_like(input_ids) if attention_mask is None else attention_mask
        emb = m.text_decoder.bert.embeddings(input_ids=input_ids)
        seq = torch.arange(length)
        causal = (seq[None, None, :].repeat(n, length, 1) <= seq[None, :, None]).to(atts.dtype)
        ext = causal[:, None, :, :] * atts[:, None, None, :]
        ext = (1.0 - ext.to(emb.dtype)) * -10000.0
        out, _ = m.text_decoder.bert.encoder(emb, attention_mask=ext, encoder_hidden_states=encoder_hidden_states,
                                             encoder_attention_mask=_ext(encoder_attention_mask, emb.dtype),
                                             return_dict=True, mode="multi_modal")
        res = rx.namespace(logits=m.text_decoder.cls(out.last_hidden_state), loss=None)
        if labels is not None:
            shifted = res.logits[:, :-1, :].contiguous()
            lab = labels[:, 1:].contiguous()
            loss = F.cross_entropy(shifted.view(-1, shifted.shape[-1]), lab.view(-1), reduction=reduction)
            res.loss = loss.view(res.logits.size(0), -1).sum(1)
        return res
    return text_decoder


def albef_vqa_case(name, case, arrays, meta):
    torch.manual_seed(0)
    ref = build_ref_albef_vqa(case)
    tied = ["text_decoder.cls.predictions.decoder.weight", "text_decoder.cls.predictions.decoder.bias"]
    listing = [e for e in ec.listing_of(ref.state_dict()) if e[0] not in tied]
    sd = ec.seeded_state_dict(listing, case["seed"])
    sd[tied[0]] = sd["text_decoder.bert.embeddings.word_embeddings.weight"]
    sd[tied[1]] = sd["text_decoder.cls.predictions.bias"]
    ref.load_state_dict(sd, strict=True)
    inp = ec.case_inputs(name, case, "albef")
    answers = ec.answer_list(case)
    fns = rx.module_items(rx.ALBEF_VQA_MODEL, ["tile"])
    methods, _ = rx.class_methods(rx.ALBEF_VQA_MODEL, "ALBEF", ["rank_answer"], extra_globals=dict(tile=fns["tile"]))
    stub = rx.make_stub(methods, text_decoder=_ref_text_decoder(ref), tokenizer=rx.namespace(pad_token_id=0))
    with torch.no_grad():
        states = _ref_question_states(ref, inp["image"], inp["ids"], inp["masks"])
        topk_ids, topk_probs = stub.rank_answer(states, inp["masks"], answers, (answers != 0).long(), case["k_test"])
        logits = stub.text_decoder(answers[:4], attention_mask=(answers[:4] != 0).long(),
                                   encoder_hidden_states=states[:1].repeat(4, 1, 1),
                                   encoder_attention_mask=inp["masks"][:1].repeat(4, 1)).logits
    arrays[name + "/question_states"] = states.numpy()
    arrays[name + "/topk_ids"] = topk_ids.numpy()
    arrays[name + "/topk_probs"] = topk_probs.numpy()
    arrays[name + "/decoder_logits"] = logits.numpy()
    meta[name] = dict(flavor="albef_vqa", listing=ec.listing_of(sd), checksums=ec.checksums(sd), seed=case["seed"], tied=tied,
                      pred=[int(topk_ids[b][int(topk_probs[b].argmax())]) for b in range(2)])


def main(npz_path=None, json_path=None):
    npz_path = npz_path or os.path.join(HERE, "encoder_golden.npz")
    json_path = json_path or os.path.join(HERE, "encoder_golden.json")
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(True)
    arrays, meta = {}, {}
    for name, case in ec.VLMO_CASES.items():
        vlmo_case(name, case, arrays, meta)
    for name, case in ec.ALBEF_CASES.items():
        albef_case(name, case, arrays, meta)
    for name, case in ec.ALBEF_VQA_CASES.items():
        albef_vqa_case(name, case, arrays, meta)
    np.savez_compressed(npz_path, **arrays)
    with open(json_path, "w") as fh:
        json.dump(meta, fh, sort_keys=True)
    return npz_path, json_path


if __name__ == "__main__":
    print(main(*sys.argv[1:3]))
