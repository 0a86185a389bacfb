"""The reference's checkpoints -> the bundled frozen white boxes (``FrozenVlmo`` / ``FrozenAlbef``).

The reference builds its models, then fills them from ``torch.load``-ed state dicts:

  * VLMo: ``VLMo.__init__`` / ``load_pretrained_weight`` (``VLMO_VQAttack/vlmo/modules/vlmo_module.py:303-324,685-804``):
    unwrap ``ckpt["state_dict" | "module" | "model"]`` (:692-705; ``module.`` prefixes of DeepSpeed checkpoints dropped,
    :114-124), cut the text position table to ``max_text_len`` (:715-733), interpolate ONE
    ``relative_position_bias_table`` of ``heads * layers`` columns geometrically when the patch grid differs (:741-804),
    ``load_state_dict(strict=False)``.  At run time the additive attention bias of layer ``l`` is
    ``table[text_imag_relative_position_index][..., l * heads:(l + 1) * heads]`` (``get_rel_pos_bias`` :806-815, index
    construction ``build_relative_position_embed`` :817-880) and the qkv bias is ``cat(q_bias, 0, v_bias)``
    (``multiway_transformer.py:92-95``).
  * ALBEF: ``adv_attack.py:83-100``: ``checkpoint['model']``, bicubic ``interpolate_pos_embed``
    (``models/vit.py:181-205``) of the ViT position table, ``load_state_dict``; the ViT's attention has ONE packed
    ``qkv`` projection (``vit.py:39``), BERT separate ``query / key / value`` (``xbert.py:224-231``).

Here the same tensors are re-laid for the forms the MI355X path consumes: the patch convolution as a GEMM weight
``(D, 3 p p)``, the packed qkv bias with its zero key part, the relative-position bias as the dense per-layer slab the
attention kernel reads in 32-key tiles, ALBEF's packed ``qkv`` split into the q / k / v the fused spec re-packs.  Nothing
is approximated: every destination element is a copy of one source element (or an exact zero), except the two
interpolations, which follow the reference's formulas.

Checked against the reference's OWN classes executed from source (``tests/golden/make_encoder_golden.py`` ->
``tests/golden/encoder_golden*.{npz,json}``; ``tests/test_reference_checkpoint.py``).
"""
import math
import re

import numpy as np
import torch
import torch.nn.functional as F


class CheckpointError(ValueError):
    """A reference state dict that does not fit the model it is loaded into."""


def extract_state_dict(ckpt):
    """The weights inside a ``torch.load``-ed checkpoint object (vlmo_module.py:692-705, :114-124; adv_attack.py:84)."""
    state = ckpt
    for key in ("state_dict", "module", "model"):
        if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict):
            state = ckpt[key]
            break
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}


def _t(sd, key):
    if key not in sd:
        raise CheckpointError("reference state dict has no {!r}".format(key))
    return sd[key].detach().to(torch.float32)


def _copy(param, value, name):
    if tuple(param.shape) != tuple(value.shape):
        raise CheckpointError("{}: checkpoint shape {} does not fit {}".format(name, tuple(value.shape), tuple(param.shape)))
    with torch.no_grad():
        param.copy_(value)


def _copy_linear(mod, sd, prefix, used):
    _copy(mod.weight, _t(sd, prefix + ".weight"), prefix + ".weight")
    _copy(mod.bias, _t(sd, prefix + ".bias"), prefix + ".bias")
    used.update((prefix + ".weight", prefix + ".bias"))


# ---- VLMo: relative position bias ------------------------------------------------------------------------------
def vlmo_relative_position_index(grid, max_text_len, max_text_len_of_initckpt=196):
    """``text_imag_relative_position_index`` (T + N, T + N) int64 and the table height it addresses
    (``build_relative_position_embed``, vlmo_module.py:817-880).  ``grid``: patches per image side."""
    num_rel = (2 * grid - 1) * (2 * grid - 1) + 3
    all_rel = num_rel + 2 * max_text_len_of_initckpt + 2
    ch, cw = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    coords = torch.stack([ch.reshape(-1), cw.reshape(-1)])                      # (2, g*g)
    rel = coords[:, :, None] - coords[:, None, :] + (grid - 1)                  # both axes shifted to start from 0
    n_img = grid * grid + 1
    image = torch.zeros(n_img, n_img, dtype=torch.int64)
    image[1:, 1:] = rel[0] * (2 * grid - 1) + rel[1]
    image[0, :] = num_rel - 3                                                   # cls -> token, token -> cls, cls -> cls
    image[:, 0] = num_rel - 2
    image[0, 0] = num_rel - 1
    pos = torch.arange(max_text_len - 1)
    text = torch.zeros(max_text_len, max_text_len, dtype=torch.int64)
    text[1:, 1:] = (pos[None, :] - pos[:, None]) - (2 - max_text_len_of_initckpt) + (num_rel + 2)
    text[0, :] = all_rel - 3
    text[:, 0] = all_rel - 2
    text[0, 0] = all_rel - 1
    top = torch.cat([text, torch.full((max_text_len, n_img), num_rel, dtype=torch.int64)], dim=1)
    bottom = torch.cat([torch.full((n_img, max_text_len), num_rel + 1, dtype=torch.int64), image], dim=1)
    return torch.cat([top, bottom], dim=0), all_rel


def interpolate_vlmo_rel_pos_table(table, dst_grid, n_extra):
    """Geometric-progression resampling of the image part of a ``relative_position_bias_table`` to another patch grid
    (vlmo_module.py:741-804, the BEiT recipe); the last ``n_extra`` rows (cls / text / cross-modal distances) are kept.

    The reference evaluates ``scipy.interpolate.interp2d(x, y, z, kind='cubic')``, which this image's SciPy (1.15) no
    longer ships; for data on a rectangular grid that class fitted ``dfitpack.regrid_smth(kx=3, ky=3, s=0)`` -- the
    routine behind ``RectBivariateSpline`` -- on ``z`` transposed, which is what is called here (SciPy's own
    ``interp2d`` transition guide).  Unpinned against the old SciPy: it cannot be installed offline."""
    from scipy.interpolate import RectBivariateSpline
    table = table.detach().to(torch.float32)
    src_num, columns = table.shape
    src_size = int(round((src_num - n_extra) ** 0.5))
    dst_size = 2 * dst_grid - 1
    if src_size * src_size != src_num - n_extra:
        raise CheckpointError("relative_position_bias_table with {} rows is not a square grid + {} extra rows"
                              .format(src_num, n_extra))
    if src_size == dst_size:
        return table
    extra, body = table[src_num - n_extra:], table[:src_num - n_extra]
    left, right = 1.01, 1.5
    while right - left > 1e-6:
        q = (left + right) / 2.0
        gp = (1.0 - q ** (src_size // 2)) / (1.0 - q)
        if gp > dst_size // 2:
            right = q
        else:
            left = q
    dis, cur = [], 1
    for i in range(src_size // 2):
        dis.append(cur)
        cur += q ** (i + 1)
    x = np.asarray([-d for d in reversed(dis)] + [0] + dis, dtype=np.float64)
    t = dst_size // 2.0
    dx = np.arange(-t, t + 0.1, 1.0)
    out = []
    for col in range(columns):
        z = body[:, col].reshape(src_size, src_size).numpy().astype(np.float64)
        f = RectBivariateSpline(x, x, z.T, kx=3, ky=3, s=0)
        out.append(torch.from_numpy(np.ascontiguousarray(f(dx, dx).T)).to(torch.float32).reshape(-1, 1))
    return torch.cat([torch.cat(out, dim=1), extra], dim=0)


def vlmo_rel_pos_bias(table, index, depth, heads):
    """Dense additive bias (depth, heads, S, S) of a ``relative_position_bias_table`` (``get_rel_pos_bias`` :806-815:
    ``F.embedding`` -> (S, S, heads * depth) -> heads-major, then one chunk of ``heads`` columns per layer)."""
    if table.shape[1] != depth * heads:
        raise CheckpointError("relative_position_bias_table has {} columns, the model needs heads * layers = {}"
                              .format(table.shape[1], depth * heads))
    if int(index.max()) >= table.shape[0]:
        raise CheckpointError("relative_position_bias_table has {} rows, the index addresses {}"
                              .format(table.shape[0], int(index.max()) + 1))
    s = index.shape[0]
    return F.embedding(index, table).permute(2, 0, 1).reshape(depth, heads, s, s).contiguous()


def vlmo_config_from_state_dict(sd, image_size, max_text_len=40, **overrides):
    """``VlmoConfig`` with the widths / depths a reference state dict implies (architecture names
    ``vlmo_base_patch16`` / ``vlmo_large_patch16``, multiway_transformer.py:385-404, are just such tuples)."""
    from .vlmo import VlmoConfig
    sd = extract_state_dict(sd)
    w = _t(sd, "transformer.patch_embed.proj.weight")
    dim, patch = int(w.shape[0]), int(w.shape[-1])
    layers = sorted({int(m.group(1)) for k in sd for m in [re.match(r"transformer\.blocks\.(\d+)\.", k)] if m})
    depth = layers[-1] + 1
    vl = [i for i in layers if "transformer.blocks.{}.mlp_vl.fc1.weight".format(i) in sd]
    table = sd.get("relative_position_bias_table")
    if table is None:
        raise CheckpointError("no relative_position_bias_table: absolute-position VLMo variants (base_plus) are outside "
                              "the attack's configurations")
    heads = int(table.shape[1]) // depth
    hidden = int(_t(sd, "transformer.blocks.0.mlp_imag.fc1.weight").shape[0])
    kw = dict(dim=dim, depth=depth, heads=heads, vlffn_start=min(vl) if vl else depth, mlp_ratio=hidden / dim, patch=patch,
              image_size=image_size, max_text_len=max_text_len,
              vocab=int(_t(sd, "text_embeddings.word_embeddings.weight").shape[0]),
              max_position=max(max_text_len, int(_t(sd, "text_embeddings.position_embeddings.weight").shape[0])),
              text_abs_pos=False, weights="reference")
    if "vqa_classifier.3.weight" in sd:
        kw["n_answers"] = int(sd["vqa_classifier.3.weight"].shape[0])
    kw.update(overrides)
    return VlmoConfig(**kw)


def load_vlmo(model, state_dict, strict=True, max_text_len_of_initckpt=None):
    """Fill a ``FrozenVlmo`` from a reference VLMo state dict.  Returns ``(missing, unexpected)`` like
    ``load_state_dict``: destination tensors nothing was found for (``strict`` raises on them -- except the VQA head,
    which pre-trained checkpoints do not carry) and checkpoint keys that were not consumed."""
    sd = extract_state_dict(state_dict)
    cfg, used, missing = model.cfg, set(), []
    d, p = cfg.dim, cfg.patch
    w = _t(sd, "transformer.patch_embed.proj.weight")
    if tuple(w.shape) != (d, 3, p, p):
        raise CheckpointError("patch embedding {} does not fit (dim {}, patch {})".format(tuple(w.shape), d, p))
    _copy(model.patch_proj.weight, w.reshape(d, 3 * p * p), "transformer.patch_embed.proj.weight")   # (c, ph, pw) order
    _copy(model.patch_proj.bias, _t(sd, "transformer.patch_embed.proj.bias"), "transformer.patch_embed.proj.bias")
    _copy(model.cls_token, _t(sd, "transformer.cls_token"), "transformer.cls_token")
    used.update(("transformer.patch_embed.proj.weight", "transformer.patch_embed.proj.bias", "transformer.cls_token"))
    if "transformer.pos_embed" in sd:
        raise CheckpointError("absolute image position embeddings (use_abs_pos_emb) are outside the attack's configurations")
    for name, ln in (("transformer.norm", model.norm),):
        _copy_linear(ln, sd, name, used)
    for i, blk in enumerate(model.blocks):
        b = "transformer.blocks.{}.".format(i)
        _copy_linear(blk.norm1, sd, b + "norm1", used)
        _copy(blk.attn.qkv.weight, _t(sd, b + "attn.qkv.weight"), b + "attn.qkv.weight")
        used.add(b + "attn.qkv.weight")
        with torch.no_grad():
            blk.attn.qkv.bias.zero_()                       # k bias is an exact zero (multiway_transformer.py:94)
            if b + "attn.q_bias" in sd:
                blk.attn.qkv.bias[:d].copy_(_t(sd, b + "attn.q_bias"))
                blk.attn.qkv.bias[2 * d:].copy_(_t(sd, b + "attn.v_bias"))
                used.update((b + "attn.q_bias", b + "attn.v_bias"))
        _copy_linear(blk.attn.proj, sd, b + "attn.proj", used)
        experts = [("norm2_text", blk.norm2_text), ("norm2_imag", blk.norm2_imag)]
        mlps = [("mlp_text", blk.mlp_text), ("mlp_imag", blk.mlp_imag)]
        if blk.mlp_vl is not None:
            experts.append(("norm2_vl", blk.norm2_vl))
            mlps.append(("mlp_vl", blk.mlp_vl))
        elif b + "mlp_vl.fc1.weight" in sd:
            raise CheckpointError("block {} of the checkpoint has a VL-FFN, the model (vlffn_start {}) has none"
                                  .format(i, cfg.vlffn_start))
        for name, ln in experts:
            _copy_linear(ln, sd, b + name, used)
        for name, mlp in mlps:
            _copy_linear(mlp.fc1, sd, b + name + ".fc1", used)
            _copy_linear(mlp.fc2, sd, b + name + ".fc2", used)
        for name, g in (("gamma_1", blk.gamma_1), ("gamma_2", blk.gamma_2)):
            if b + name in sd:
                _copy(g, _t(sd, b + name), b + name)
                used.add(b + name)
            else:                                           # layer_scale_init_values=None: gamma is the float 1.0
                with torch.no_grad():
                    g.fill_(1.0)
    # text side: BertEmbeddings (transformers==4.8.1, V/requirements.txt:1) -- with position_embedding_type "rel_pos"
    # (vlmo_module.py:216) the position table is loaded but never added
    te = "text_embeddings."
    _copy(model.word_embeddings.weight, _t(sd, te + "word_embeddings.weight"), te + "word_embeddings.weight")
    pos = _t(sd, te + "position_embeddings.weight")
    with torch.no_grad():
        model.position_embeddings.weight.zero_()
        rows = min(pos.shape[0], model.position_embeddings.weight.shape[0])
        model.position_embeddings.weight[:rows].copy_(pos[:rows])
    _copy(model.bert_type_embeddings.weight, _t(sd, te + "token_type_embeddings.weight"), te + "token_type_embeddings.weight")
    _copy_linear(model.bert_ln, sd, te + "LayerNorm", used)
    used.update((te + "word_embeddings.weight", te + "position_embeddings.weight", te + "token_type_embeddings.weight",
                 te + "position_ids"))
    _copy(model.token_type_embeddings.weight, _t(sd, "token_type_embeddings.weight")[:2], "token_type_embeddings.weight")
    used.add("token_type_embeddings.weight")
    _copy_linear(model.pooler, sd, "pooler.dense", used)
    if "mlm_score.decoder.weight" in sd:                    # MLMHead(bert_config): the decoder is NOT tied (heads.py:40-47)
        _copy_linear(model.mlm_dense, sd, "mlm_score.transform.dense", used)
        _copy_linear(model.mlm_ln, sd, "mlm_score.transform.LayerNorm", used)
        _copy(model.mlm_bias, _t(sd, "mlm_score.bias"), "mlm_score.bias")
        model.set_mlm_decoder(_t(sd, "mlm_score.decoder.weight"))
        used.update(("mlm_score.bias", "mlm_score.decoder.weight"))
    else:
        missing.append("mlm_score")
    if model.vqa_classifier is not None:
        if "vqa_classifier.0.weight" in sd:
            for j in (0, 1, 3):
                _copy_linear(model.vqa_classifier[j], sd, "vqa_classifier.{}".format(j), used)
        else:
            missing.append("vqa_classifier")
    # relative position bias
    grid = cfg.image_size // p
    table = _t(sd, "relative_position_bias_table")
    used.add("relative_position_bias_table")
    if max_text_len_of_initckpt is None:                    # config.py:39 -- no named config of the reference changes it
        max_text_len_of_initckpt = 196
    index, all_rel = vlmo_relative_position_index(grid, cfg.max_text_len, max_text_len_of_initckpt)
    if table.shape[0] != all_rel:                           # another patch grid: vlmo_module.py:749-803
        n_extra = all_rel - (2 * grid - 1) ** 2
        table = interpolate_vlmo_rel_pos_table(table, grid, n_extra)
    bias = vlmo_rel_pos_bias(table, index, cfg.depth, cfg.heads)
    if tuple(bias.shape) != tuple(model.rel_pos_bias.shape):
        raise CheckpointError("relative position bias {} does not fit the model's {}".format(tuple(bias.shape),
                                                                                        tuple(model.rel_pos_bias.shape)))
    with torch.no_grad():
        model.rel_pos_bias.copy_(bias)
    model.invalidate_fused()
    unexpected = sorted(k for k in sd if k not in used and not k.endswith("relative_position_index"))
    hard = [m for m in missing if m != "vqa_classifier"]
    if strict and hard:
        raise CheckpointError("reference state dict lacks: {}".format(", ".join(hard)))
    return missing, unexpected


def vlmo_from_reference(ckpt, image_size=384, max_text_len=40, vqa_head=None, strict=True, **overrides):
    """``FrozenVlmo`` built for and filled from a reference checkpoint object / state dict."""
    from .vlmo import FrozenVlmo
    sd = extract_state_dict(ckpt)
    cfg = vlmo_config_from_state_dict(sd, image_size, max_text_len, **overrides)
    head = ("vqa_classifier.0.weight" in sd) if vqa_head is None else vqa_head
    model = FrozenVlmo(cfg, seed=0, vqa_head=head)
    load_vlmo(model, sd, strict=strict)
    return model


# ---- ALBEF ----------------------------------------------------------------------------------------------------------
def interpolate_vit_pos_embed(pos_embed, n_tokens_dst):
    """Bicubic resampling of a ViT position table (1, 1 + g*g, D) to another grid; the class token's row is kept
    (``interpolate_pos_embed``, ALBEF_attack/models/vit.py:181-205)."""
    pos_embed = pos_embed.detach().to(torch.float32)
    d = pos_embed.shape[-1]
    n_src, n_dst = pos_embed.shape[-2] - 1, n_tokens_dst - 1
    src, dst = int(n_src ** 0.5), int(n_dst ** 0.5)
    if src == dst:
        return pos_embed
    grid = pos_embed[:, 1:].reshape(-1, src, src, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(dst, dst), mode="bicubic", align_corners=False)
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).flatten(1, 2)], dim=1)


def _load_bert_layer(layer, sd, b, used, dim):
    """One ``xbert.BertLayer`` (xbert.py:438-520 and its parts :218-436) -> ``albef._BertLayer``."""
    for src, dst in (("query", layer.attn.q), ("key", layer.attn.k), ("value", layer.attn.v)):
        _copy_linear(dst, sd, b + "attention.self." + src, used)
    _copy_linear(layer.attn.o, sd, b + "attention.output.dense", used)
    _copy_linear(layer.ln_attn, sd, b + "attention.output.LayerNorm", used)
    if layer.cross is not None:
        for src, dst in (("query", layer.cross.q), ("key", layer.cross.k), ("value", layer.cross.v)):
            _copy_linear(dst, sd, b + "crossattention.self." + src, used)
        _copy_linear(layer.cross.o, sd, b + "crossattention.output.dense", used)
        _copy_linear(layer.ln_cross, sd, b + "crossattention.output.LayerNorm", used)
    elif b + "crossattention.self.query.weight" in sd:
        raise CheckpointError("{} has cross-attention, the model's layer has none (fusion_layer)".format(b))
    _copy_linear(layer.mlp.fc1, sd, b + "intermediate.dense", used)
    _copy_linear(layer.mlp.fc2, sd, b + "output.dense", used)
    _copy_linear(layer.ln_out, sd, b + "output.LayerNorm", used)


def load_albef(model, state_dict, strict=True):
    """Fill a ``FrozenAlbef`` from a reference state dict: the pre-trained model (``ALBEF_pre``: ``text_encoder.bert.*``
    + ``text_encoder.cls.*``; adv_attack.py:83-92) or the VQA model (``ALBEF``: ``text_encoder.*`` +
    ``text_decoder.*``; :96-100) -- told apart by their key prefixes.  Momentum copies (``*_m``), queues and the
    ITC / ITM heads are not on the attack path and are reported as unexpected."""
    sd = extract_state_dict(state_dict)
    cfg, used, missing = model.cfg, set(), []
    d, p = cfg.dim, cfg.patch
    v = "visual_encoder."
    w = _t(sd, v + "patch_embed.proj.weight")
    if tuple(w.shape) != (d, 3, p, p):
        raise CheckpointError("patch embedding {} does not fit (dim {}, patch {})".format(tuple(w.shape), d, p))
    _copy(model.patch_proj.weight, w.reshape(d, 3 * p * p), v + "patch_embed.proj.weight")
    _copy(model.patch_proj.bias, _t(sd, v + "patch_embed.proj.bias"), v + "patch_embed.proj.bias")
    _copy(model.cls_token, _t(sd, v + "cls_token"), v + "cls_token")
    _copy(model.pos_embed, interpolate_vit_pos_embed(_t(sd, v + "pos_embed"), cfg.n_image_tokens), v + "pos_embed")
    used.update((v + "patch_embed.proj.weight", v + "patch_embed.proj.bias", v + "cls_token", v + "pos_embed"))
    for i, blk in enumerate(model.vit_blocks):
        b = v + "blocks.{}.".format(i)
        _copy_linear(blk.norm1, sd, b + "norm1", used)
        qkv_w, qkv_b = _t(sd, b + "attn.qkv.weight"), _t(sd, b + "attn.qkv.bias")
        if tuple(qkv_w.shape) != (3 * d, d):
            raise CheckpointError("{}attn.qkv.weight {} does not fit dim {}".format(b, tuple(qkv_w.shape), d))
        for j, lin in enumerate((blk.attn.q, blk.attn.k, blk.attn.v)):     # reshape(B, N, 3, H, d): q rows first
            _copy(lin.weight, qkv_w[j * d:(j + 1) * d], b + "attn.qkv.weight")
            _copy(lin.bias, qkv_b[j * d:(j + 1) * d], b + "attn.qkv.bias")
        used.update((b + "attn.qkv.weight", b + "attn.qkv.bias"))
        _copy_linear(blk.attn.o, sd, b + "attn.proj", used)
        _copy_linear(blk.norm2, sd, b + "norm2", used)
        _copy_linear(blk.mlp.fc1, sd, b + "mlp.fc1", used)
        _copy_linear(blk.mlp.fc2, sd, b + "mlp.fc2", used)
    _copy_linear(model.vit_norm, sd, v + "norm", used)
    t = "text_encoder.bert." if "text_encoder.bert.embeddings.word_embeddings.weight" in sd else "text_encoder."
    e = t + "embeddings."
    _copy(model.word_embeddings.weight, _t(sd, e + "word_embeddings.weight"), e + "word_embeddings.weight")
    pos = _t(sd, e + "position_embeddings.weight")
    with torch.no_grad():
        model.position_embeddings.weight.zero_()
        rows = min(pos.shape[0], model.position_embeddings.weight.shape[0])
        model.position_embeddings.weight[:rows].copy_(pos[:rows])
    _copy(model.type_embeddings.weight, _t(sd, e + "token_type_embeddings.weight"), e + "token_type_embeddings.weight")
    _copy_linear(model.emb_ln, sd, e + "LayerNorm", used)
    used.update((e + "word_embeddings.weight", e + "position_embeddings.weight", e + "token_type_embeddings.weight",
                 e + "position_ids"))
    for i, layer in enumerate(model.bert_layers):
        _load_bert_layer(layer, sd, t + "encoder.layer.{}.".format(i), used, d)
    c = "text_encoder.cls.predictions."
    if c + "transform.dense.weight" in sd:                  # BertForMaskedLM: decoder tied to the word embeddings
        _copy_linear(model.mlm_dense, sd, c + "transform.dense", used)
        _copy_linear(model.mlm_ln, sd, c + "transform.LayerNorm", used)
        _copy(model.mlm_bias, _t(sd, c + "bias"), c + "bias")
        used.update((c + "bias", c + "decoder.weight", c + "decoder.bias"))
    else:
        missing.append("text_encoder.cls")
    if model.has_vqa:
        dd = "text_decoder.bert."
        if dd + "embeddings.word_embeddings.weight" in sd:
            de = dd + "embeddings."
            _copy(model.dec_word.weight, _t(sd, de + "word_embeddings.weight"), de + "word_embeddings.weight")
            pos = _t(sd, de + "position_embeddings.weight")
            with torch.no_grad():
                model.dec_pos.weight.zero_()
                rows = min(pos.shape[0], model.dec_pos.weight.shape[0])
                model.dec_pos.weight[:rows].copy_(pos[:rows])
            _copy(model.dec_type.weight, _t(sd, de + "token_type_embeddings.weight"), de + "token_type_embeddings.weight")
            _copy_linear(model.dec_emb_ln, sd, de + "LayerNorm", used)
            used.update((de + "word_embeddings.weight", de + "position_embeddings.weight",
                         de + "token_type_embeddings.weight", de + "position_ids"))
            for i, layer in enumerate(model.dec_layers):
                _load_bert_layer(layer, sd, dd + "encoder.layer.{}.".format(i), used, d)
            dc = "text_decoder.cls.predictions."
            _copy_linear(model.dec_dense, sd, dc + "transform.dense", used)
            _copy_linear(model.dec_ln, sd, dc + "transform.LayerNorm", used)
            _copy(model.dec_bias, _t(sd, dc + "bias"), dc + "bias")
            used.update((dc + "bias", dc + "decoder.weight", dc + "decoder.bias"))
        else:
            missing.append("text_decoder")
    model.invalidate_fused()
    unexpected = sorted(k for k in sd if k not in used)
    hard = [m for m in missing if not (m == "text_encoder.cls" and model.has_vqa)]
    if strict and hard:
        raise CheckpointError("reference state dict lacks: {}".format(", ".join(hard)))
    return missing, unexpected


def albef_config_from_state_dict(sd, image_size, **overrides):
    from .albef import AlbefConfig
    sd = extract_state_dict(sd)
    w = _t(sd, "visual_encoder.patch_embed.proj.weight")
    dim, patch = int(w.shape[0]), int(w.shape[-1])

    def depth_of(pattern):
        idx = [int(m.group(1)) for k in sd for m in [re.match(pattern, k)] if m]
        return max(idx) + 1 if idx else 0

    t = r"text_encoder\.bert\." if "text_encoder.bert.embeddings.word_embeddings.weight" in sd else r"text_encoder\."
    bert_depth = depth_of(t + r"encoder\.layer\.(\d+)\.")
    cross = [int(m.group(1)) for k in sd for m in [re.match(t + r"encoder\.layer\.(\d+)\.crossattention\.", k)] if m]
    e = ("text_encoder.bert." if "bert" in t else "text_encoder.") + "embeddings."
    kw = dict(dim=dim, patch=patch, image_size=image_size, vit_depth=depth_of(r"visual_encoder\.blocks\.(\d+)\."),
              bert_depth=bert_depth, fusion_layer=min(cross) if cross else bert_depth,
              vocab=int(_t(sd, e + "word_embeddings.weight").shape[0]),
              max_position=int(_t(sd, e + "position_embeddings.weight").shape[0]), weights="reference")
    dec = depth_of(r"text_decoder\.bert\.encoder\.layer\.(\d+)\.")
    if dec:
        kw["decoder_depth"] = dec
    kw.update(overrides)
    if "heads" not in kw:
        kw["heads"] = max(1, dim // 64)
    return AlbefConfig(**kw)


def albef_from_reference(ckpt, image_size=384, vqa_head=None, strict=True, **overrides):
    from .albef import FrozenAlbef
    sd = extract_state_dict(ckpt)
    cfg = albef_config_from_state_dict(sd, image_size, **overrides)
    head = ("text_decoder.bert.embeddings.word_embeddings.weight" in sd) if vqa_head is None else vqa_head
    model = FrozenAlbef(cfg, seed=0, vqa_head=head)
    load_albef(model, sd, strict=strict)
    return model
