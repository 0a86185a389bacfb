"""The bundled white boxes, filled from REFERENCE-format state dicts, reproduce the reference's own model classes.

Fixtures: ``tests/golden/encoder_golden.{npz,json}``, produced in the build container by ``make_encoder_golden.py``, which
executes the reference's ``Mlp`` / ``Attention`` / ``Block`` / ``MultiWayTransformer`` (multiway_transformer.py:33-383),
``VLMo.build_relative_position_embed`` / ``get_rel_pos_bias`` / ``pgd_attack`` / ``pgd_attack_vl`` / ``pgd_mlm_attack``
(vlmo_module.py:806-880, :1328-1529), ALBEF's ``VisionTransformer`` (models/vit.py) and ``BertEmbeddings`` ...
``BertEncoder`` + MLM head (models/xbert.py:169-700) straight from their source files.  The state dicts carry the reference's
key names (``transformer.blocks.0.attn.q_bias``, ``relative_position_bias_table``,
``text_encoder.bert.encoder.layer.1.crossattention.self.query.weight`` ...) and go through
``whitebox/checkpoint.py`` -- the loader a user with a real checkpoint calls.

Tolerances (fp32): feature maps / logits 1e-4 of the map's largest magnitude; image and text-embedding gradients 1e-3
relative (l2 over the stored elements).  ``-m "not gpu"``: the host path of the same modules (eager block loop, torch
attention).  ``-m gpu``: csrc/attn.hip + csrc/block.hip + library GEMMs (the fused encoder for head size 64: the
``head64`` and base-width cases at 591 / 915 tokens; the eager loop over attn.hip with zero-padded heads for ``tiny``).
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden import encoder_cases as ec
from vqattack_amd.whitebox import checkpoint as ck
from vqattack_amd.whitebox.albef import AlbefAttackAdapters
from vqattack_amd.whitebox.vlmo import VlmoAttackAdapters

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FEAT_TOL, GRAD_TOL = 1e-4, 1e-3
_ERRORS = {}          # what -> largest error seen; written to $VQA_PARITY_REPORT (a profiles/ record) when that is set


def _note(what, err):
    _ERRORS[what] = max(_ERRORS.get(what, 0.0), err)


@pytest.fixture(scope="module", autouse=True)
def _parity_report():
    yield
    path = os.environ.get("VQA_PARITY_REPORT")
    if path and _ERRORS:
        worst = {}
        for what, err in _ERRORS.items():                       # "case/sN/quantity" -> per case and quantity
            parts = what.split("/")
            key = parts[0] + "/" + parts[-1] if len(parts) > 1 else what
            worst[key] = max(worst.get(key, 0.0), err)
        with open(path, "a") as fh:
            fh.write(json.dumps({"device": "cuda" if torch.cuda.is_available() else "cpu", "feat_tol": FEAT_TOL,
                                 "grad_tol": GRAD_TOL, "largest_error": worst}, sort_keys=True) + "\n")


@pytest.fixture(scope="module")
def enc():
    with open(os.path.join(GOLD, "encoder_golden.json")) as fh:
        meta = json.load(fh)
    return np.load(os.path.join(GOLD, "encoder_golden.npz")), meta


def _state_dict(enc, name):
    """The case's reference-format state dict: regenerated from (listing, seed), verified against the recorded
    checksums, and -- where the fixture carries the weights -- against those bit for bit."""
    z, meta = enc
    rec = meta[name]
    tied = set(rec.get("tied", []))
    sd = ec.seeded_state_dict([e for e in rec["listing"] if e[0] not in tied], rec["seed"])
    for key in tied:                             # the MLM / LM decoders are tied to their word embeddings and output bias
        stem = key.split(".cls.")[0]
        emb = stem + (".bert" if stem + ".bert.embeddings.word_embeddings.weight" in sd else "")
        sd[key] = sd[emb + ".embeddings.word_embeddings.weight"] if key.endswith("weight") else \
            sd[stem + ".cls.predictions.bias"]
    for k, want in rec["checksums"].items():
        assert ec.checksum(sd[k]) == want, k
    stored = [k for k in z.files if k.startswith(name + "/sd/")]
    for k in stored:
        assert np.array_equal(z[k], sd[k[len(name) + 4:]].numpy()), k
    return sd


def _close(got, want, tol, what):
    want = torch.as_tensor(np.asarray(want), dtype=torch.float32)
    got = got.detach().float().cpu()
    assert tuple(got.shape) == tuple(want.shape), (what, tuple(got.shape), tuple(want.shape))
    scale = float(want.abs().max()) or 1.0
    err = float((got - want).abs().max()) / scale
    _note(what, err)
    assert err <= tol, "{}: max error {:.3e} of the largest magnitude (limit {:.0e})".format(what, err, tol)


def _rel(got, want, tol, what):
    want = torch.as_tensor(np.asarray(want), dtype=torch.float64)
    got = got.detach().double().cpu()
    assert tuple(got.shape) == tuple(want.shape), (what, tuple(got.shape), tuple(want.shape))
    err = float((got - want).norm() / want.norm().clamp_min(1e-30))
    _note(what, err)
    assert err <= tol, "{}: relative l2 error {:.3e} (limit {:.0e})".format(what, err, tol)


# ---- VLMo ----------------------------------------------------------------------------------------------------------
def _vlmo_model(enc, name, device):
    case = ec.VLMO_CASES[name]
    sd = _state_dict(enc, name)
    model = ck.vlmo_from_reference({"state_dict": sd}, image_size=case["image_size"], max_text_len=case["max_text_len"])
    assert model.cfg.dim == case["dim"] and model.cfg.depth == case["depth"] and model.cfg.heads == case["heads"]
    assert model.cfg.vlffn_start == case["vlffn_start"] and model.cfg.n_answers == case["n_answers"]
    assert model.vqa_classifier is not None and not model.cfg.text_abs_pos
    return model.to(device), case


def _vlmo_scalar(out, weights, n_text, tlen):
    """sum_b <[pooler(cls), cls per layer, real text + image tokens per layer]_b, W_b>: the product-side form of the
    generator's functional over the reference's packed outputs (vlmo_module.py:1435-1444)."""
    pooled, _, lf = out
    total = 0.0
    for b, (w0, w1, w2) in enumerate(weights):
        maps = torch.stack([t[b] for t in lf.layers])                               # (depth + 1, S, D)
        tokens = torch.cat([maps[:, :n_text[b]], maps[:, tlen:]], dim=1)
        total = total + (pooled[b:b + 1] * w0).sum() + (maps[:, 0] * w1).sum() + (tokens * w2).sum()
    return total


def _check_vlmo(enc, name, device):
    z, meta = enc
    model, case = _vlmo_model(enc, name, device)
    inp = {k: v.to(device) for k, v in ec.case_inputs(name, case, "vlmo").items()}
    n_text = list(case["text_lens"])
    ad = VlmoAttackAdapters(model)
    ad.set_text(inp["ids"], inp["masks"], text_ids_mlm=inp["mlm_ids"], text_mask_mlm=inp["masks"])
    tlen = ad._tlen
    assert tlen == max(n_text)                                                      # trailing padding is not encoded
    d, depth = case["dim"], case["depth"]
    n_img = (case["image_size"] // case["patch"]) ** 2 + 1
    weights = []
    for b in range(2):
        rows = n_text[b] + n_img
        weights.append([w.to(device) for w in ec.functional_weights([(1, d), (depth + 1, d), (depth + 1, rows, d)],
                                                                    case["seed"] + b)])
    x = inp["image"].clone().requires_grad_(True)
    with torch.enable_grad():
        out = ad.pgd_attack(x)
        _vlmo_scalar(out, weights, n_text, tlen).backward()
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        maps = torch.stack([t[b] for t in out[2].layers]).detach()
        tokens = torch.cat([maps[:, :n_text[b]], maps[:, tlen:]], dim=1)
        assert tokens.shape[1] == meta[name]["samples"][b]["n_rows"]
        rows = torch.as_tensor(z[key + "feats_rows"]).to(device)
        _close(tokens[:, rows], z[key + "feats"], FEAT_TOL, key + "feats")
        _rel(tokens.double().pow(2).sum(dim=(1, 2)).sqrt(), z[key + "feats_norm"], FEAT_TOL, key + "feats_norm")
        _close(maps[:, 0], z[key + "cls_per_layer"], FEAT_TOL, key + "cls_per_layer")
        _close(out[0][b:b + 1], z[key + "cls_feats"], FEAT_TOL, key + "cls_feats")
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x.grad[b].reshape(-1)[flat], z[key + "grad_image"], GRAD_TOL, key + "grad_image")
        _rel(x.grad[b].double().norm().reshape(()), z[key + "grad_image_norm"], GRAD_TOL, key + "grad_image_norm")
    # text embeddings ("rel_pos": no absolute position table), then the image + text-embedding closure
    emb = model.text_embeddings(inp["ids"])
    for b in range(2):
        _close(emb[b], z["{}/s{}/text_embeds".format(name, b)], FEAT_TOL, name + "/text_embeds")
    x2, e2 = inp["image"].clone().requires_grad_(True), emb.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        _vlmo_scalar(ad.pgd_attack_vl([x2, e2]), weights, n_text, tlen).backward()
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x2.grad[b].reshape(-1)[flat], z[key + "vl_grad_image"], GRAD_TOL, key + "vl_grad_image")
        _rel(e2.grad[b], z[key + "vl_grad_text"], GRAD_TOL, key + "vl_grad_text")
    # MLM closure (dense contract) and the VQA head
    x3 = inp["image"].clone().requires_grad_(True)
    with torch.enable_grad():
        logits = ad.pgd_mlm_attack(x3)[0]
        total = 0.0
        for b in range(2):
            w = ec.functional_weights([(1, n_text[b], case["vocab"])], case["seed"] + 50 + b)[0].to(device)
            total = total + (logits[b:b + 1, :n_text[b]] * w).sum()
        total.backward()
    assert logits.shape[1] == case["max_text_len"]                                  # the reference's dense contract
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        _close(logits[b, :n_text[b]], z[key + "mlm_logits"], FEAT_TOL, key + "mlm_logits")
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x3.grad[b].reshape(-1)[flat], z[key + "mlm_grad_image"], GRAD_TOL, key + "mlm_grad_image")
    with torch.no_grad():
        _, states = model.encode(inp["image"], model.text_embeddings(inp["ids"]), inp["masks"])
        vqa = model.vqa_classifier(model.pooled(states))
    for b in range(2):
        _close(vqa[b:b + 1], z["{}/s{}/vqa_logits".format(name, b)], FEAT_TOL, name + "/vqa_logits")
    return model


@pytest.mark.parametrize("name", list(ec.VLMO_CASES))
def test_vlmo_reference_state_dict_host(enc, name):
    _check_vlmo(enc, name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ec.VLMO_CASES))
def test_vlmo_reference_state_dict_hip(enc, name):
    from vqattack_amd.whitebox import _fused
    case = ec.VLMO_CASES[name]
    model = _check_vlmo(enc, name, "cuda")
    # which device path produced the numbers: the graph-free fused encoder for head size 64, else the eager block loop
    # over csrc/attn.hip with zero-padded heads
    assert _fused.supported(model.cfg.dim, model.cfg.heads) == (case["dim"] // case["heads"] == 64)
    assert (model._fused_spec is not None) == (case["dim"] // case["heads"] == 64)


def test_vlmo_relative_position_index_and_bias_are_the_references(enc):
    z, meta = enc
    for name, case in ec.VLMO_CASES.items():
        grid = case["image_size"] // case["patch"]
        index, all_rel = ck.vlmo_relative_position_index(grid, case["max_text_len"], 196)
        table = _state_dict(enc, name)["relative_position_bias_table"]
        assert table.shape[0] == all_rel
        rows = z[name + "/rel_pos_rows"]
        ref_index = z[name + "/rel_pos_index"]
        if ref_index.shape[0] == index.shape[0]:
            assert np.array_equal(index.numpy(), ref_index)
        else:
            assert np.array_equal(index[rows].numpy(), ref_index)
        bias = ck.vlmo_rel_pos_bias(table, index, case["depth"], case["heads"])
        want = z[name + "/rel_pos_bias_rows"]
        got = bias if want.shape[2] == bias.shape[2] else bias[:, :, rows]
        assert np.array_equal(got.numpy(), want), name                              # a gather: bit-exact


def test_vlmo_loader_reports_and_refuses(enc):
    sd = _state_dict(enc, "vlmo_tiny")
    case = ec.VLMO_CASES["vlmo_tiny"]
    model = ck.vlmo_from_reference(sd, image_size=case["image_size"])
    extra = dict(sd, **{"itm_score.fc.weight": torch.zeros(2, 64), "module.logit_scale": torch.zeros(())})
    missing, unexpected = model.load_reference_state_dict(extra)
    assert missing == [] and unexpected == ["itm_score.fc.weight", "logit_scale"]
    # the key bias is an exact zero, q / v carry the checkpoint's vectors
    d = case["dim"]
    qkv_b = model.blocks[0].attn.qkv.bias
    assert torch.equal(qkv_b[:d], sd["transformer.blocks.0.attn.q_bias"]) and not bool(qkv_b[d:2 * d].any())
    assert torch.equal(qkv_b[2 * d:], sd["transformer.blocks.0.attn.v_bias"])
    assert model.mlm_decoder is not None and torch.equal(model.mlm_decoder, sd["mlm_score.decoder.weight"])
    # a pre-trained checkpoint (no VQA head) loads into a white box, and into a black box only when not strict
    pre = {k: v for k, v in sd.items() if not k.startswith("vqa_classifier")}
    white = ck.vlmo_from_reference(pre, image_size=case["image_size"])
    assert white.vqa_classifier is None
    no_mlm = {k: v for k, v in sd.items() if not k.startswith("mlm_score")}
    with pytest.raises(ck.CheckpointError):
        ck.vlmo_from_reference(no_mlm, image_size=case["image_size"])
    bad = dict(sd)
    bad["transformer.blocks.1.attn.proj.weight"] = torch.zeros(3, 3)
    with pytest.raises(ck.CheckpointError):
        ck.vlmo_from_reference(bad, image_size=case["image_size"])
    with pytest.raises(ck.CheckpointError):                                       # a table of another depth
        model.load_reference_state_dict(dict(sd, relative_position_bias_table=torch.zeros(446, 8)))


def test_vlmo_rel_pos_table_is_resampled_for_another_resolution(enc):
    """vlmo_module.py:741-804: a checkpoint of a 4 x 4 patch grid loaded into an 8 x 8 model keeps its class / text /
    cross-modal rows, resamples the (2g - 1)^2 image distances and reproduces the source values where the geometric
    source positions coincide with target positions (centre, and the outermost ring: the progression ends on it)."""
    sd = _state_dict(enc, "vlmo_tiny")
    case = ec.VLMO_CASES["vlmo_tiny"]
    table = sd["relative_position_bias_table"]
    n_extra = 3 + 2 * 196 + 2
    out = ck.interpolate_vlmo_rel_pos_table(table, 8, n_extra)
    assert out.shape == ((2 * 8 - 1) ** 2 + n_extra, table.shape[1])
    assert torch.equal(out[-n_extra:], table[-n_extra:])
    src, dst = table[:-n_extra].reshape(7, 7, -1), out[:-n_extra].reshape(15, 15, -1)
    assert torch.allclose(dst[7, 7], src[3, 3], atol=1e-5)
    assert torch.allclose(dst[0, 0], src[0, 0], atol=1e-3) and torch.allclose(dst[14, 14], src[6, 6], atol=1e-3)
    model = ck.vlmo_from_reference(sd, image_size=2 * case["image_size"])            # the loader takes that path itself
    assert model.rel_pos_bias.shape[-1] == case["max_text_len"] + 8 * 8 + 1
    index, _ = ck.vlmo_relative_position_index(8, case["max_text_len"], 196)
    assert torch.equal(model.rel_pos_bias, ck.vlmo_rel_pos_bias(out, index, case["depth"], case["heads"]))


# ---- ALBEF ---------------------------------------------------------------------------------------------------------
def _albef_model(enc, name, device):
    case = ec.ALBEF_CASES[name]
    sd = _state_dict(enc, name)
    model = ck.albef_from_reference({"model": sd}, image_size=case["image_size"], heads=case["heads"], mlm_probability=0.0)
    c = model.cfg
    assert (c.dim, c.vit_depth, c.bert_depth, c.fusion_layer) == (case["dim"], case["vit_depth"], case["bert_depth"],
                                                                  case["fusion_layer"])
    return model.to(device), case


def _albef_scalar(out, weights, n_text):
    txt, img = out
    total = 0.0
    for b, (wt, wi) in enumerate(weights):
        t = torch.stack([m[b] for m in txt.layers])[:, :n_text[b]]
        i = torch.stack([m[b] for m in img.layers])
        total = total + (t * wt).sum() + (i * wi).sum()
    return total


def _check_albef(enc, name, device):
    z, _ = enc
    model, case = _albef_model(enc, name, device)
    inp = {k: v.to(device) for k, v in ec.case_inputs(name, case, "albef").items()}
    n_text = list(case["text_lens"])
    d = case["dim"]
    n_img = (case["image_size"] // case["patch"]) ** 2 + 1
    ad = AlbefAttackAdapters(model)
    ad.set_text(inp["ids"], inp["masks"], text_ids_mlm=inp["mlm_ids"], text_mask_mlm=inp["masks"])
    weights = [[w.to(device) for w in ec.functional_weights([(case["bert_depth"] + 1, n_text[b], d),
                                                             (case["vit_depth"] + 1, n_img, d)], case["seed"] + b)]
               for b in range(2)]
    x = inp["image"].clone().requires_grad_(True)
    with torch.enable_grad():
        out = ad.pgd_attack(x)
        _albef_scalar(out, weights, n_text).backward()
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        img = torch.stack([m[b] for m in out[1].layers]).detach()
        txt = torch.stack([m[b] for m in out[0].layers]).detach()[:, :n_text[b]]
        rows = torch.as_tensor(z[key + "img_rows"]).to(device)
        _close(img[:, rows], z[key + "img_feats"], FEAT_TOL, key + "img_feats")
        _rel(img.double().pow(2).sum(dim=(1, 2)).sqrt(), z[key + "img_feats_norm"], FEAT_TOL, key + "img_feats_norm")
        _close(txt, z[key + "txt_feats"], FEAT_TOL, key + "txt_feats")
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x.grad[b].reshape(-1)[flat], z[key + "grad_image"], GRAD_TOL, key + "grad_image")
        _rel(x.grad[b].double().norm().reshape(()), z[key + "grad_image_norm"], GRAD_TOL, key + "grad_image_norm")
    emb = model.text_embeddings(inp["ids"])
    for b in range(2):
        _close(emb[b], z["{}/s{}/text_embeds".format(name, b)], FEAT_TOL, name + "/text_embeds")
    x2, e2 = inp["image"].clone().requires_grad_(True), emb.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        _albef_scalar(ad.pgd_attack_vl([x2, e2]), weights, n_text).backward()
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x2.grad[b].reshape(-1)[flat], z[key + "vl_grad_image"], GRAD_TOL, key + "vl_grad_image")
        _rel(e2.grad[b, :n_text[b]], z[key + "vl_grad_text"][:n_text[b]], GRAD_TOL, key + "vl_grad_text")
    x3 = inp["image"].clone().requires_grad_(True)
    with torch.enable_grad():
        logits = ad.pgd_mlm_attack(x3)[0]
        total = 0.0
        for b in range(2):
            w = ec.functional_weights([(1, n_text[b], case["vocab"])], case["seed"] + 50 + b)[0].to(device)
            total = total + (logits[b:b + 1, :n_text[b]] * w).sum()
        total.backward()
    for b in range(2):
        key = "{}/s{}/".format(name, b)
        _close(logits[b, :n_text[b]], z[key + "mlm_logits"], FEAT_TOL, key + "mlm_logits")
        flat = torch.as_tensor(z[key + "grad_index"].astype(np.int64)).to(device)
        _rel(x3.grad[b].reshape(-1)[flat], z[key + "mlm_grad_image"], GRAD_TOL, key + "mlm_grad_image")


@pytest.mark.parametrize("name", list(ec.ALBEF_CASES))
def test_albef_reference_state_dict_host(enc, name):
    _check_albef(enc, name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ec.ALBEF_CASES))
def test_albef_reference_state_dict_hip(enc, name):
    _check_albef(enc, name, "cuda")


def test_albef_position_table_resampling_is_the_references(enc):
    z, _ = enc
    for name in ec.ALBEF_CASES:
        sd = _state_dict(enc, name)
        got = ck.interpolate_vit_pos_embed(sd["visual_encoder.pos_embed"], int(z[name + "/pos_embed_dst_tokens"]))
        rows = torch.as_tensor(z[name + "/pos_embed_rows"])
        assert np.array_equal(got[0, rows].numpy(), z[name + "/pos_embed_resampled"]), name


def test_albef_loader_splits_qkv_and_reports(enc):
    sd = _state_dict(enc, "albef_tiny")
    case = ec.ALBEF_CASES["albef_tiny"]
    model = ck.albef_from_reference(sd, image_size=case["image_size"], heads=case["heads"])
    d = case["dim"]
    w = sd["visual_encoder.blocks.1.attn.qkv.weight"]
    a = model.vit_blocks[1].attn
    assert torch.equal(a.q.weight, w[:d]) and torch.equal(a.k.weight, w[d:2 * d]) and torch.equal(a.v.weight, w[2 * d:])
    lay = model.bert_layers[1]
    assert lay.cross is not None and model.bert_layers[0].cross is None
    assert torch.equal(lay.cross.k.weight, sd["text_encoder.bert.encoder.layer.1.crossattention.self.key.weight"])
    extra = dict(sd, **{"visual_encoder_m.cls_token": torch.zeros(1, 1, d), "temp": torch.ones(())})
    missing, unexpected = model.load_reference_state_dict(extra)
    assert missing == [] and unexpected == ["temp", "visual_encoder_m.cls_token"]
    with pytest.raises(ck.CheckpointError):
        model.load_reference_state_dict({k: v for k, v in sd.items() if "layer.1.crossattention" not in k})
    # a checkpoint at another resolution: the position table is resampled on the way in
    bigger = ck.albef_from_reference(sd, image_size=2 * case["image_size"], heads=case["heads"])
    assert bigger.pos_embed.shape[1] == (2 * case["image_size"] // case["patch"]) ** 2 + 1
    assert torch.equal(bigger.pos_embed[:, 0], sd["visual_encoder.pos_embed"][:, 0])


# ---- ALBEF VQA victim (black box): fusion encoder + answer decoder + rank_answer ------------------------------------
def _check_albef_vqa(enc, name, device):
    """The fine-tuned checkpoint's key layout (``text_encoder.*``, ``text_decoder.bert.*``, ``text_decoder.cls.*``) loaded
    into the bundled victim: question states, decoder logits and the re-ranked top-k answers equal what the reference's
    ``BertEncoder`` / ``BertOnlyMLMHead`` classes and its ``rank_answer`` (model_vqa.py:149-203) produce."""
    z, meta = enc
    case = ec.ALBEF_VQA_CASES[name]
    sd = _state_dict(enc, name)
    model = ck.albef_from_reference({"model": sd}, image_size=case["image_size"], heads=case["heads"], mlm_probability=0.0,
                                    k_test=case["k_test"]).to(device)
    assert model.has_vqa and model.cfg.decoder_depth == case["dec_depth"] and model.cfg.fusion_layer == case["fusion_layer"]
    answers = ec.answer_list(case).to(device)
    model.set_answer_list(answers)
    inp = {k: v.to(device) for k, v in ec.case_inputs(name, case, "albef").items()}
    with torch.no_grad():
        image_states, _ = model.visual_encoder(inp["image"])
        states, _ = model.text_encoder(model.text_embeddings(inp["ids"]), inp["masks"], image_states)
        for b, n in enumerate(case["text_lens"]):                         # padded query rows are not comparable
            _close(states[b, :n], z[name + "/question_states"][b, :n], FEAT_TOL, name + "/question_states")
        logits = model._decode(answers[:4], (answers[:4] != 0).long(), states[:1].repeat(4, 1, 1), inp["masks"][:1].repeat(4, 1))
        real = (answers[:4] != 0).cpu()
        want = torch.as_tensor(z[name + "/decoder_logits"])
        err = float((logits.cpu() - want)[real].abs().max()) / float(want[real].abs().max())
        _note(name + "/decoder_logits", err)
        assert err <= FEAT_TOL, err
        topk_ids, topk_probs = model.rank_answer(states, inp["masks"])
        pred = model.vqa_answer(inp["image"], inp["ids"], inp["masks"])
    assert pred.cpu().tolist() == meta[name]["pred"]
    # the reference returns the k candidates re-ranked; same set, same order, same probabilities
    ref_ids, ref_probs = torch.as_tensor(z[name + "/topk_ids"]), torch.as_tensor(z[name + "/topk_probs"])
    assert torch.equal(topk_ids.cpu(), ref_ids)
    assert torch.allclose(topk_probs.cpu(), ref_probs, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", list(ec.ALBEF_VQA_CASES))
def test_albef_vqa_victim_reference_state_dict_host(enc, name):
    _check_albef_vqa(enc, name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ec.ALBEF_VQA_CASES))
def test_albef_vqa_victim_reference_state_dict_hip(enc, name):
    _check_albef_vqa(enc, name, "cuda")
