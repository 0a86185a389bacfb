"""Second half of ``make_text_golden.py``: MLM-task construction, the full per-sample attack loops, VLMo feature
packing and ``rank_answer`` -- all executed from the reference source (see ``refexec.py``).  Build container only."""
import copy
import importlib
import sys
import types

import numpy as np
import torch

from tests.golden import refexec as rx
from tests.golden import textworld as tw
from tests.golden.make_text_golden import (A_TEXT_METHODS, V_TEXT_METHODS, albef_stub, ref_embeddings, vlmo_stub,
                                           word_info)

CLEVERHANS = {"albef": rx.REF + "/ALBEF_VQAttack/cleverhans", "vlmo": rx.REF + "/VLMO_VQAttack/cleverhans"}

# (vilt answer, all correct answers, white-box answer, paraphrase, question)
TASK_CASES = [
    ("red", ["red"], "red", "the umbrella is red.", "what color is the umbrella"),
    ("blue", ["blue", "green", "dark blue"], "blue", "the color of the kite is blue.", "what color is the kite"),
    ("two", ["two", "three", "2"], "two", "there are two cats on the table.", "how many cats are on the table"),
    ("zebra", ["zebra", "horse"], "zebra", "the animal in the field is a giraffe.", "which animal is in the field"),
    ("playing frisbee", ["playing frisbee", "flying kite", "eating", "playing ball"], "playing frisbee",
     "the man is playing frisbee in the field.", "what is the man playing"),
    ("on table", ["on table", "in bowl", "table"], "on table", "the cats are sitting on the table.",
     "where are the cats sitting"),
    ("cats", ["cats", "dogs", "zebras"], "cats", "the animals on the couch are cats.", "what animals are on the couch"),
    ("cat", ["cat"], "cat", "the cat sees the other cat.", "what does the cat see"),
]


def _load_cleverhans(flavor):
    """The reference's own cleverhans copy (importable once an empty ``torchvision`` placeholder exists: see
    tests/golden/make_golden.py)."""
    for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
        del sys.modules[name]
    if "torchvision" not in sys.modules:
        tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    sys.path[:] = [p for p in sys.path if p not in CLEVERHANS.values()]
    sys.path.insert(0, CLEVERHANS[flavor])
    pgd = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent")
    pgd_vl = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent_vl")
    assert pgd.__file__.startswith(CLEVERHANS[flavor])
    return pgd, pgd_vl


class _Recorder:
    """Wraps a reference cleverhans module: same call, but every (adv, second return value) is recorded."""

    def __init__(self, module):
        self._m = module
        self.calls = []

    def projected_gradient_descent(self, *a, **kw):
        adv, second = self._m.projected_gradient_descent(*a, **kw)
        self.calls.append((kw.get("ls"), a[4] if len(a) > 4 else kw.get("nb_iter"), second))
        return adv, second


class _CaptureUniform:
    """Records the tensors ``Tensor.uniform_`` fills while active (the ``time == 0`` start of the reference's PGD)."""

    def __enter__(self):
        self.drawn = []
        self._orig = torch.Tensor.uniform_
        outer = self

        def rec(t, *a, **kw):
            out = outer._orig(t, *a, **kw)
            outer.drawn.append(out.detach().clone())
            return out
        torch.Tensor.uniform_ = rec
        return self

    def __exit__(self, *exc):
        torch.Tensor.uniform_ = self._orig
        return False


# ------------------------------------------------------------------------------------------- MLM task construction
def task_inputs(world, vilt_ans, all_ans, para):
    """Id-level inputs of the MLM-task construction: words as word-piece id lists."""
    return dict(ans_word_ids=word_info(world, vilt_ans.lower())["word_ids"],
                all_ans_word_ids=[word_info(world, a.lower())["word_ids"] for a in all_ans],
                same_as_vilt=[a == vilt_ans for a in all_ans],
                pa_word_ids=word_info(world, para.strip(".").lower())["word_ids"],
                stop_word_ids=[word_info(world, w)["word_ids"][0] for w in ("on", "and", "in", "his", "her", "its")],
                period_id=int(world.tok.convert_tokens_to_ids(".")))


def _task_block(flavor):
    if flavor == "albef":
        return rx.method_block(rx.ALBEF_ATTACK, "Adv_attack", "evaluate", 428, 558, ["batch", "ret"],
                               "attack_batch = copy.deepcopy(batch)",
                               returns=["old_alg", "attack_batch", "batch"]), "question_id"
    return rx.method_block(rx.VLMO_MODULE, "VLMo", "test_step", 1743, 1889, ["batch", "ret"],
                           "attack_batch = copy.deepcopy(batch)",
                           returns=["old_alg", "attack_batch", "batch"]), "qid"


def section_tasks(world, arrays, meta):
    """old_alg + [MASK]-ed paraphrase + MLM labels for hand-built (answer, paraphrase) pairs, both copies."""
    out = []
    for flavor, mk in (("albef", albef_stub), ("vlmo", vlmo_stub)):
        block, qkey = _task_block(flavor)
        for ci, (vilt_ans, all_ans, ans, para, question) in enumerate(TASK_CASES):
            stub, _ = mk(world, 64)
            qid = 1000 + ci
            stub.vilt_ans_table = {str(qid): vilt_ans}
            stub.all_correct_ans = {str(qid): list(all_ans)}
            stub.chatgpt = {str(qid): [question, para]}
            batch = {qkey: [qid], "question": [question], "text": [question + "?"],
                     "image": torch.zeros(1, 3, 8, 8), "text_labels": torch.full((1, 40), -100, dtype=torch.long)}
            ret = {"preds": [ans]}
            with rx.cpu_as_cuda(), torch.no_grad():
                try:
                    res = block(stub, batch, ret)
                    err = None
                except UnboundLocalError as exc:        # the reference leaves mlm_labels unbound for an empty answer set
                    res, err = None, str(exc)
            entry = dict(flavor=flavor, vilt_ans=vilt_ans, all_correct_ans=list(all_ans), ans=ans, paraphrase=para,
                         question=question, error=err, **task_inputs(world, vilt_ans, all_ans, para))
            if res is not None:
                # the block's own locals of interest (mlm_labels is only bound on the old_alg == 0 path)
                entry["old_alg"] = int(res["old_alg"])
                if res["old_alg"] == 0:
                    entry["text_ids_mlm"] = res["attack_batch"]["text_ids_mlm"].tolist()
                    entry["text_mask_mlm"] = res["attack_batch"]["text_mask_mlm"].tolist()
                    entry["text_labels_mlm"] = res["batch"]["text_labels_mlm"].tolist()
            out.append(entry)
    meta["task_cases"] = out
    # mlm_labels (2-d or 3-d) is a local of the block: a second pass returning it where it is bound
    labelled = []
    for flavor, mk in (("albef", albef_stub), ("vlmo", vlmo_stub)):
        if flavor == "albef":
            block = rx.method_block(rx.ALBEF_ATTACK, "Adv_attack", "evaluate", 428, 558, ["batch", "ret"],
                                    "attack_batch = copy.deepcopy(batch)", returns=["mlm_labels", "list_words"])
            qkey = "question_id"
        else:
            block = rx.method_block(rx.VLMO_MODULE, "VLMo", "test_step", 1743, 1889, ["batch", "ret"],
                                    "attack_batch = copy.deepcopy(batch)", returns=["mlm_labels", "list_words"])
            qkey = "qid"
        for ci, (vilt_ans, all_ans, ans, para, question) in enumerate(TASK_CASES):
            entry = next(e for e in out if e["flavor"] == flavor and e["paraphrase"] == para and e["vilt_ans"] == vilt_ans)
            if entry.get("old_alg") != 0:
                continue
            stub, _ = mk(world, 64)
            qid = 1000 + ci
            stub.vilt_ans_table = {str(qid): vilt_ans}
            stub.all_correct_ans = {str(qid): list(all_ans)}
            stub.chatgpt = {str(qid): [question, para]}
            batch = {qkey: [qid], "question": [question], "text": [question + "?"],
                     "image": torch.zeros(1, 3, 8, 8), "text_labels": torch.full((1, 40), -100, dtype=torch.long)}
            with rx.cpu_as_cuda(), torch.no_grad():
                res = block(stub, batch, {"preds": [ans]})
            entry["mlm_labels"] = res["mlm_labels"].tolist()
            entry["list_words"] = res["list_words"]
            labelled.append((flavor, ci))
    meta["task_labelled"] = labelled


# ------------------------------------------------------------------------------------------- full ALBEF loop
class _AlbefWhite:
    """The tiny frozen ALBEF of this repository under the reference's method names (model_pretrain.py:85-141)."""

    def __init__(self, model):
        self.m = model

    def Gen_feats(self, image, ids, masks):
        return self.m.gen_feats(image, ids, masks)

    def Gen_feats_from_embeds(self, image, embeds, ids, masks):
        return self.m.gen_feats_from_embeds(image, embeds, ids, masks)

    def get_mlm_logits(self, image, ids, masks):
        return self.m.get_mlm_logits(image, ids, masks)


LOOP_CASES = [
    # (name, question, vilt answer, all correct answers, paraphrase, seed)
    ("feat_words", "red table cat man umbrella playing", "zebra", ["zebra"], "the animal is a giraffe.", 11),
    ("dual_words", "is the man holding a red umbrella", "red", ["red", "blue"], "the umbrella is red.", 12),
    ("feat_nowords", "is the", "zebra", ["zebra"], "the animal is a giraffe.", 13),
    ("dual_nowords", "is the", "red", ["red"], "the umbrella is red.", 14),
    # multi-piece words ("cats" = cat ##s, not substitutable) next to substitutable ones
    ("feat_pieces", "how many cats are playing on the table", "zebra", ["zebra"], "the animal is a giraffe.", 15),
    # dual loss with THREE label sets (3-d labels): two alternative answers of the same piece count
    ("dual_3d_labels", "what color is the kite", "blue", ["blue", "green", "red"], "the color of the kite is blue.", 16),
]


def section_albef_loop(world, arrays, meta):
    """adv_attack.py:428-712 end to end on a tiny white box: reference control flow, reference cleverhans operators,
    reference text update -- only the encoder inside the model_fn closures is this repository's."""
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_tiny
    pgd_mod, pgd_vl_mod = _load_cleverhans("albef")
    cfg = albef_tiny(mlm_probability=0.0)
    model = FrozenAlbef(cfg, seed=3)
    tabs = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in model.embedding_tables().items()}
    cases = []
    for name, question, vilt_ans, all_ans, para, seed in LOOP_CASES:
        rec, rec_vl = _Recorder(pgd_mod), _Recorder(pgd_vl_mod)
        block = rx.method_block(rx.ALBEF_ATTACK, "Adv_attack", "evaluate", 428, 712, ["batch", "ret"],
                                "attack_batch = copy.deepcopy(batch)", extra_globals=dict(pgd=rec, pgd_vl=rec_vl),
                                returns=["adv_x", "adv_text", "old_alg", "iter_list"])
        methods, ns = rx.class_methods(rx.ALBEF_ATTACK, "Adv_attack", A_TEXT_METHODS,
                                       extra_globals=dict(filter_words=world.filter_words))
        ns["Feature"] = rx.module_items(rx.ALBEF_ATTACK, ["Feature"])["Feature"]
        stub = rx.make_stub(methods, tokenizer_mlm=world.tok, tokenizer=world.tok, mlm_model=world.mlm,
                            USE_model=world.use, device=torch.device("cpu"), embeddings=ref_embeddings(tabs), batch={},
                            white_model=_AlbefWhite(model), vilt_ans_table={"7": vilt_ans},
                            all_correct_ans={"7": list(all_ans)}, chatgpt={"7": [question, para]})
        g = torch.Generator().manual_seed(seed)
        image = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
        batch = {"question_id": [7], "question": [question], "image": image.clone()}
        torch.manual_seed(seed)
        with rx.cpu_as_cuda(), _CaptureUniform() as cap:
            res = block(stub, batch, {"preds": ["x"]})
        torch.set_grad_enabled(True)
        eta = [t for t in cap.drawn if t.shape == image.shape]
        assert len(eta) == 1, "expected exactly one random start"
        key = "loop_albef_" + name
        arrays[key + "_image"] = image.numpy()
        arrays[key + "_eta"] = eta[0].numpy()
        arrays[key + "_adv"] = res["adv_x"].detach().numpy()
        enc = world.tok(question, padding="longest", truncation=True, max_length=25)
        adv_enc = world.tok(res["adv_text"], padding="longest", truncation=True, max_length=25)
        cases.append(dict(name=name, key=key, question=question, vilt_ans=vilt_ans, all_correct_ans=list(all_ans),
                          paraphrase=para, old_alg=int(res["old_alg"]), iter_list=res["iter_list"],
                          info=word_info(world, question), task=task_inputs(world, vilt_ans, all_ans, para),
                          text_ids=enc["input_ids"], adv_text=res["adv_text"], adv_text_ids=adv_enc["input_ids"],
                          pgd_calls=[dict(ls=c[0], nb_iter=c[1], losses=[float(v) for v in c[2]]) for c in rec.calls],
                          n_probe_calls=len(rec_vl.calls)))
    meta["loop_albef"] = dict(cases=cases, model=dict(config="albef_tiny(mlm_probability=0.0)", seed=3,
                                                      weight_checksum=_checksum(model)))


def _checksum(model):
    s = 0.0
    for p in model.parameters():
        s += float(p.detach().double().abs().sum())
    return s


# ------------------------------------------------------------------------------------------- VLMo packing + loop
class _VlmoShim:
    """This repository's tiny frozen VLMo behind the attribute names the reference's VLMo methods use
    (vlmo_module.py:1328-1529: ``transformer.visual_embed / blocks / norm``, ``token_type_embeddings``,
    ``get_rel_pos_bias``, ``pooler``, ``mlm_score``, ``text_embeddings``, ``self(batch)``)."""

    def __init__(self, model):
        m = self.m = model
        self.token_type_embeddings = m.token_type_embeddings
        self.text_imag_relative_position_index = None
        self.num_layers = m.cfg.depth

        class _Blk:
            def __init__(s, blk):
                s.blk = blk

            def __call__(s, x, mask=None, modality_type=None, relative_position_bias=None):
                pad = torch.zeros(x.shape[0], 1, 1, x.shape[1]).masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
                return s.blk(x, relative_position_bias.unsqueeze(0) + pad, m.cfg.max_text_len)

        self.transformer = rx.namespace(
            visual_embed=lambda x: (m.visual_embed(x), torch.ones(x.shape[0], m.cfg.n_image_tokens)),
            blocks=[_Blk(b) for b in m.blocks], norm=m.norm)

    def get_rel_pos_bias(self, _index):
        return [self.m.rel_pos_bias[i] for i in range(self.m.cfg.depth)]

    def pooler(self, x):
        return self.m.pooled(x)

    def mlm_score(self, text_feats):
        return self.m.mlm_score(text_feats)

    def text_embeddings(self, ids):
        return self.m.text_embeddings(ids)

    def infer(self, batch):
        m = self.m
        feats, _ = m.encode(batch["image"][0], m.text_embeddings(batch["text_ids"]), batch["text_masks"])
        return {"feats_list": feats, "text_masks": batch["text_masks"]}


V_MODEL_METHODS = ["Gen_ori_feats", "pgd_attack", "pgd_attack_vl", "pgd_mlm_attack"]


def _vlmo_full_stub(world, model, **attrs):
    shim = _VlmoShim(model)
    methods, ns = rx.class_methods(rx.VLMO_MODULE, "VLMo", V_TEXT_METHODS + V_MODEL_METHODS,
                                   extra_globals=dict(filter_words=world.filter_words,
                                                      vlmo_utils=rx.namespace(set_task=lambda _s: None)))
    ns["Feature"] = rx.module_items(rx.VLMO_MODULE, ["Feature"])["Feature"]
    methods = dict(methods)
    methods["__call__"] = lambda self, batch: shim.infer(batch)
    base = dict(tokenizer_mlm=world.tok, mlm_model=world.mlm, USE_model=world.use, device=torch.device("cpu"), batch={},
                transformer=shim.transformer, token_type_embeddings=shim.token_type_embeddings,
                text_imag_relative_position_index=None, get_rel_pos_bias=shim.get_rel_pos_bias, pooler=shim.pooler,
                mlm_score=shim.mlm_score, text_embeddings=shim.text_embeddings)
    base.update(attrs)
    return rx.make_stub(methods, **base)


def vlmo_tiny40():
    from vqattack_amd.whitebox.vlmo import VlmoConfig
    # the reference hard-codes the text length 40 in its packing (vlmo_module.py:1306,1379,1440)
    return VlmoConfig(dim=64, depth=3, heads=4, vlffn_start=2, image_size=32, patch=8, max_text_len=40, n_answers=17)


def section_vlmo_pack(world, arrays, meta):
    from vqattack_amd.whitebox.vlmo import FrozenVlmo
    model = FrozenVlmo(vlmo_tiny40(), seed=5)
    stub = _vlmo_full_stub(world, model)
    g = torch.Generator().manual_seed(31)
    image = torch.empty(1, 3, 32, 32).uniform_(-1, 1, generator=g)
    question = "is the man holding a red umbrella?"
    enc = world.tok(question, padding="max_length", truncation=True, max_length=40)
    ids, masks = torch.tensor(enc["input_ids"]).unsqueeze(0), torch.tensor(enc["attention_mask"]).unsqueeze(0)
    mlm_ids = ids.clone()
    mlm_ids[0, 3] = tw.MASK
    stub.batch = {"text_ids": ids, "text_masks": masks, "text_ids_mlm": mlm_ids, "text_mask_mlm": masks}
    with rx.cpu_as_cuda(), torch.no_grad():
        out = stub.pgd_attack(image)
        out_vl = stub.pgd_attack_vl([image, model.text_embeddings(ids)])
        out_mlm = stub.pgd_mlm_attack(image)
        ori = stub.Gen_ori_feats({"image": [image], "text_ids": ids, "text_masks": masks})
    arrays["pack_vlmo_image"] = image.numpy()
    for tag, o in (("pgd", out), ("vl", out_vl), ("mlm", out_mlm), ("ori", ori)):
        for i, t in enumerate(o):
            t = t.detach()
            if t.shape[-1] == 30522:           # MLM logits: keep a 256-column slice + the per-row sums (fixture size)
                arrays["pack_vlmo_{}_{}_rowsum".format(tag, i)] = t.double().sum(-1).numpy()
                t = t[..., :256]
            arrays["pack_vlmo_{}_{}".format(tag, i)] = t.numpy()
    meta["pack_vlmo"] = dict(question=question, text_ids=ids.tolist(), text_masks=masks.tolist(),
                             text_ids_mlm=mlm_ids.tolist(), model=dict(config="vlmo_tiny40", seed=5,
                                                                       weight_checksum=_checksum(model)))


def section_vlmo_loop(world, arrays, meta):
    """vlmo_module.py:1743-2057 end to end (test_step's attack part) on the tiny VLMo."""
    from vqattack_amd.whitebox.vlmo import FrozenVlmo
    pgd_mod, pgd_vl_mod = _load_cleverhans("vlmo")
    model = FrozenVlmo(vlmo_tiny40(), seed=5)
    cases = []
    for name, question, vilt_ans, all_ans, para, seed in LOOP_CASES:
        rec, rec_vl = _Recorder(pgd_mod), _Recorder(pgd_vl_mod)
        block = rx.method_block(rx.VLMO_MODULE, "VLMo", "test_step", 1743, 2057, ["batch", "ret"],
                                "attack_batch = copy.deepcopy(batch)", extra_globals=dict(pgd=rec, pgd_vl=rec_vl),
                                returns=["adv_x", "adv_text", "old_alg", "iter_list"])
        stub = _vlmo_full_stub(world, model, vilt_ans_table={"7": vilt_ans}, all_correct_ans={"7": list(all_ans)},
                               chatgpt={"7": [question, para]})
        g = torch.Generator().manual_seed(seed)
        image = torch.empty(1, 3, 32, 32).uniform_(-1, 1, generator=g)
        text = question + "?"
        enc = world.tok(text, padding="max_length", truncation=True, max_length=40)
        ids, masks = torch.tensor(enc["input_ids"]).unsqueeze(0), torch.tensor(enc["attention_mask"]).unsqueeze(0)
        batch = {"qid": [7], "text": [text], "image": [image.clone()], "text_ids": ids, "text_masks": masks,
                 "text_labels": torch.full((1, 40), -100, dtype=torch.long)}
        torch.manual_seed(seed)
        with rx.cpu_as_cuda(), _CaptureUniform() as cap:
            res = block(stub, batch, {"preds": ["x"]})
        torch.set_grad_enabled(True)
        eta = [t for t in cap.drawn if t.shape == image.shape]
        assert len(eta) == 1
        key = "loop_vlmo_" + name
        arrays[key + "_image"] = image.numpy()
        arrays[key + "_eta"] = eta[0].numpy()
        arrays[key + "_adv"] = res["adv_x"].detach().numpy()
        adv_enc = world.tok(res["adv_text"], padding="max_length", truncation=True, max_length=40)
        cases.append(dict(name=name, key=key, question=text, vilt_ans=vilt_ans, all_correct_ans=list(all_ans),
                          paraphrase=para, old_alg=int(res["old_alg"]), iter_list=res["iter_list"],
                          info=word_info(world, question), task=task_inputs(world, vilt_ans, all_ans, para),
                          text_ids=enc["input_ids"], text_masks=enc["attention_mask"], adv_text=res["adv_text"],
                          adv_text_ids=adv_enc["input_ids"],
                          pgd_calls=[dict(ls=c[0], nb_iter=c[1], losses=[float(v) for v in c[2]]) for c in rec.calls],
                          n_probe_calls=len(rec_vl.calls)))
    meta["loop_vlmo"] = dict(cases=cases, model=dict(config="vlmo_tiny40", seed=5, weight_checksum=_checksum(model)))


# ------------------------------------------------------------------------------------------- rank_answer
def section_rank_answer(world, arrays, meta):
    """model_vqa.py:149-203 on the tiny ALBEF victim's decoder: per-question-list loop, k_test re-ranking."""
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_tiny
    import torch.nn.functional as F
    cfg = albef_tiny(mlm_probability=0.0)
    white = FrozenAlbef(cfg, seed=3)
    black = FrozenAlbef.finetuned_from(white, seed=4)
    fns = rx.module_items(rx.ALBEF_VQA_MODEL, ["tile"])
    methods, _ = rx.class_methods(rx.ALBEF_VQA_MODEL, "ALBEF", ["rank_answer"], extra_globals=dict(tile=fns["tile"]))

    class _Out:
        pass

    def text_decoder(input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                     labels=None, return_dict=True, reduction="none"):
        # BertLMHeadModel (xbert.py:1265-1271) around this repository's decoder trunk: shifted per-token CE, summed
        atts = torch.ones_like(input_ids) if attention_mask is None else attention_mask
        logits = black._decode(input_ids, atts, encoder_hidden_states, encoder_attention_mask)
        out = _Out()
        out.logits = logits
        if labels is not None:
            shifted = logits[:, :-1, :].contiguous()
            lab = labels[:, 1:].contiguous()
            loss = F.cross_entropy(shifted.view(-1, shifted.shape[-1]), lab.view(-1), reduction="none")
            out.loss = loss.view(logits.size(0), -1).sum(1)
        return out

    stub = rx.make_stub(methods, text_decoder=text_decoder, tokenizer=rx.namespace(pad_token_id=cfg.pad_id))
    g = torch.Generator().manual_seed(77)
    images = torch.empty(6, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    ids = torch.zeros(6, 8, dtype=torch.long)
    for b in range(6):
        n = 3 + b % 4
        ids[b, 0] = 101
        ids[b, 1:1 + n] = torch.randint(1000, 30522, (n,), generator=g)
        ids[b, 1 + n] = 102
    masks = (ids != 0).long()
    with torch.no_grad():
        image_states, _ = black.visual_encoder(images)
        states, _ = black.text_encoder(black.text_embeddings(ids), masks, image_states)
        ans = black.answer_ids
        topk_ids, topk_probs = stub.rank_answer(states, masks, ans, (ans != cfg.pad_id).long(), cfg.k_test)
        # what adv_attack.py:722-726 then reads: the best re-ranked answer of each question
        pred = [int(topk_ids[b][int(topk_probs[b].argmax())]) for b in range(6)]
    arrays["rank_images"] = images.numpy()
    arrays["rank_topk_ids"] = topk_ids.numpy()
    arrays["rank_topk_probs"] = topk_probs.numpy()
    meta["rank"] = dict(text_ids=ids.tolist(), pred=pred, k=cfg.k_test,
                        model=dict(white_seed=3, black_seed=4, weight_checksum=_checksum(black)))


def run(world, arrays, meta):
    section_tasks(world, arrays, meta)
    section_albef_loop(world, arrays, meta)
    section_vlmo_pack(world, arrays, meta)
    section_vlmo_loop(world, arrays, meta)
    section_rank_answer(world, arrays, meta)
