"""Generate ``tests/golden/text_golden.npz`` + ``text_golden.json``: reference-produced vectors for the TEXT side of
the joint attack (rows a9-a11 of SURVEY.md section 8a) and for the black-box scorer (f2).

Build container only (needs ``/root/reference``).  The reference's orchestrators cannot be imported (tensorflow_hub,
nltk, timm, pytorch_lightning ... are missing), so their hot-path methods are compiled from the reference source with
``ast`` and EXECUTED as they stand (``tests/golden/refexec.py``) against stub objects built from this repository's
synthetic text world (``tests/golden/textworld.py``): a Hugging Face ``BertTokenizer`` over a small vocabulary, seeded
embedding tables, a table-lookup MLM and a bag-of-embeddings sentence encoder in place of BERT-MLM / TF-Hub USE.
Only inputs and outputs are stored; no reference source or bytecode enters the repository.

    PYTHONDONTWRITEBYTECODE=1 python -m tests.golden.make_text_golden

Sections (fixture key prefixes):
  emb_*      BertEmbeddings (ALBEF_attack/models/xbert.py:169-216) + Adv_attack.text_embeddings (adv_attack.py:369-384)
  dirsim_*   dir_sim (adv_attack.py:325-333; vlmo_module.py:1632-1640)
  sched      iter_list of cal_text_attack_list for 0..40 substitutable words (adv_attack.py:229-239; vlmo :1545-1556)
  cand_*     cal_text_attack_list + get_substitues (adv_attack.py:191-264; vlmo_module.py:1531-1630)
  upd_*      update_adv_text (adv_attack.py:265-324; vlmo_module.py:1642-1702) and update_mlm_text (:334-351)
  task_*     old_alg decision + [MASK]-ed paraphrase + MLM labels (adv_attack.py:428-558; vlmo_module.py:1743-1891)
  loop_*     the whole per-sample attack (adv_attack.py:428-712) over a tiny white box, reference cleverhans operators
  pack_*     VLMo feature packing (vlmo_module.py:1287-1312, 1328-1446)
  rank_*     rank_answer (ALBEF_attack/models/model_vqa.py:149-203)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden import refexec as rx  # noqa: E402
from tests.golden import textworld as tw  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_NPZ = os.path.join(HERE, "text_golden.npz")
OUT_JSON = os.path.join(HERE, "text_golden.json")

A_TEXT_METHODS = ["_tokenize", "filter", "get_substitues", "get_bpe_substitues", "cal_text_attack_list",
                  "update_adv_text", "dir_sim", "update_mlm_text", "text_embeddings", "pgd_attack", "pgd_attack_vl",
                  "pgd_mlm_attack", "Gen_ori_feats"]
V_TEXT_METHODS = ["_tokenize", "filter", "get_substitues", "get_bpe_substitues", "cal_text_attack_list",
                  "update_adv_text", "dir_sim", "update_mlm_text"]


class World:
    """Everything the stubs need, built once."""

    def __init__(self):
        self.vocab = tw.build_vocab()
        self.V = len(self.vocab)
        self.tok = tw.make_tokenizer(self.vocab)
        self.filter_words = [w for w in rx.load_filter_words()]
        self.filter_in_vocab = sorted(w for w in set(self.filter_words) if w in set(self.vocab))
        self.use_table = tw.seeded(21, self.V, 24)
        mlm = tw.seeded(22, self.V, self.V, scale=0.16)
        drift = tw.seeded(23, 64, self.V, scale=0.04)
        idx = {w: i for i, w in enumerate(self.vocab)}
        # special / [unusedN] entries never rank: as STRINGS they do not re-tokenise to themselves ("[unused7]" splits at
        # the brackets), which is outside the id-level restatement's stated assumptions (oracle/text_scoring.py)
        mlm[:, :tw.MASK + 1] = -5.0
        # hand-placed boosts: original word on top, a word piece and a stop word among the best, strong candidates
        for w, tops in (("cat", ["cat", "dog", "##s", "the", "horse", "zebra"]), ("red", ["blue", "red", "green", "on"]),
                        ("table", ["desk", "couch", "##ing", "chair"]), ("umbrella", ["kite", "umbrella"]),
                        ("man", ["woman", "man", "people"]), ("playing", ["holding", "flying", "eating"])):
            for rank, t in enumerate(tops):
                mlm[idx[w], idx[t]] = 1.2 - 0.12 * rank
        self.mlm = tw.MlmStandIn(mlm, drift)
        self.use = tw.SentenceEncoderStandIn(self.tok, self.use_table)
        self.tables = {d: tw.embedding_tables(self.V, d, seed=7 + d) for d in (64, 768)}


def ref_embeddings(tables):
    """The reference's own BertEmbeddings class (xbert.py:169-216) carrying the fixture tables, eval mode."""
    cls = rx.module_items(rx.ALBEF_XBERT, ["BertEmbeddings"])["BertEmbeddings"]
    v, d = tables["word"].shape
    cfg = rx.namespace(vocab_size=v, hidden_size=d, pad_token_id=0, max_position_embeddings=tables["pos"].shape[0],
                       type_vocab_size=2, layer_norm_eps=tables["ln_eps"], hidden_dropout_prob=0.0)
    emb = cls(cfg).eval()
    with torch.no_grad():
        emb.word_embeddings.weight.copy_(torch.from_numpy(tables["word"]))
        emb.position_embeddings.weight.copy_(torch.from_numpy(tables["pos"]))
        emb.token_type_embeddings.weight.copy_(torch.from_numpy(tables["type_emb"]))
        emb.LayerNorm.weight.copy_(torch.from_numpy(tables["gamma"]))
        emb.LayerNorm.bias.copy_(torch.from_numpy(tables["beta"]))
    return emb


def albef_stub(world, dim=64, **attrs):
    methods, ns = rx.class_methods(rx.ALBEF_ATTACK, "Adv_attack", A_TEXT_METHODS,
                                   extra_globals=dict(filter_words=world.filter_words))
    ns["Feature"] = rx.module_items(rx.ALBEF_ATTACK, ["Feature"])["Feature"]
    emb = ref_embeddings(world.tables[dim])
    base = dict(tokenizer_mlm=world.tok, tokenizer=world.tok, mlm_model=world.mlm, USE_model=world.use,
                device=torch.device("cpu"), embeddings=emb, batch={})
    base.update(attrs)
    return rx.make_stub(methods, **base), ns


def vlmo_stub(world, dim=64, **attrs):
    methods, ns = rx.class_methods(rx.VLMO_MODULE, "VLMo", V_TEXT_METHODS,
                                   extra_globals=dict(filter_words=world.filter_words,
                                                      vlmo_utils=rx.namespace(set_task=lambda _self: None)))
    ns["Feature"] = rx.module_items(rx.VLMO_MODULE, ["Feature"])["Feature"]
    emb = ref_embeddings(world.tables[dim])
    base = dict(tokenizer_mlm=world.tok, mlm_model=world.mlm, USE_model=world.use, device=torch.device("cpu"),
                batch={}, text_embeddings=lambda ids: emb(input_ids=ids))
    base.update(attrs)
    return rx.make_stub(methods, **base), ns


# ----------------------------------------------------------------------------------------------- sections
def section_embeddings(world, arrays, meta):
    """ids -> BertEmbeddings output, through Adv_attack.text_embeddings, for both table sizes."""
    g = np.random.RandomState(5)
    for dim in (64, 768):
        stub, _ = albef_stub(world, dim)
        ids = g.randint(0, world.V, size=(3, 11)).astype(np.int64)
        ids[0, :3] = [tw.CLS, tw.MASK, tw.SEP]
        ids[1, -4:] = tw.PAD
        with torch.no_grad():
            out = stub.text_embeddings(torch.from_numpy(ids))
        # cross-check with Hugging Face's BertEmbeddings (the third-party class xbert.py vendors)
        from transformers import BertConfig
        from transformers.models.bert.modeling_bert import BertEmbeddings as HFEmb
        t = world.tables[dim]
        hf = HFEmb(BertConfig(vocab_size=world.V, hidden_size=dim, max_position_embeddings=t["pos"].shape[0],
                              layer_norm_eps=t["ln_eps"], hidden_dropout_prob=0.0, num_attention_heads=4)).eval()
        hf.load_state_dict(stub.embeddings.state_dict(), strict=False)
        with torch.no_grad():
            hf_out = hf(input_ids=torch.from_numpy(ids))
        assert torch.equal(out, hf_out), "reference BertEmbeddings != Hugging Face BertEmbeddings"
        arrays["emb_ids_{}".format(dim)] = ids
        arrays["emb_out_{}".format(dim)] = out.numpy()
        for k, v in t.items():
            if k != "ln_eps":
                arrays["tab{}_{}".format(dim, k)] = v
    meta["ln_eps"] = 1e-12


def section_dir_sim(world, arrays, meta):
    g = np.random.RandomState(6)
    a = g.standard_normal((12, 64)).astype(np.float32)
    b = g.standard_normal((12, 64)).astype(np.float32)
    a[3] = 0.0                        # zero direction: F.normalize clamps the norm at 1e-12
    b[4] = 0.0
    a[5] = b[5]                       # identical directions
    a[6] = -b[6]
    a[7] *= 1e-9                      # tiny but non-zero
    sa, _ = albef_stub(world)
    sv, _ = vlmo_stub(world)
    out_a = np.array([float(sa.dir_sim(torch.from_numpy(x), torch.from_numpy(y))) for x, y in zip(a, b)], np.float32)
    out_v = np.array([float(sv.dir_sim(torch.from_numpy(x), torch.from_numpy(y))) for x, y in zip(a, b)], np.float32)
    assert np.array_equal(out_a, out_v)
    arrays["dirsim_a"], arrays["dirsim_b"], arrays["dirsim_out"] = a, b, out_a


def _words(world, n):
    """n distinct single-piece, non-stop words of the vocabulary."""
    pool = [w for w in tw.WORDS if w not in set(world.filter_words)]
    assert len(pool) >= n, (len(pool), n)
    return pool[:n]


def section_schedule(world, arrays, meta):
    """iter_list for 0 .. 40 substitutable words, from cal_text_attack_list itself (both copies)."""
    sa, _ = albef_stub(world)
    sv, _ = vlmo_stub(world)
    rows = []
    for n in range(0, 41):
        sent = " ".join(_words(world, n)) if n else "the"       # "the" is a stop word: no substitutable word
        it_a, _ = sa.cal_text_attack_list(sent)
        it_v, _ = sv.cal_text_attack_list(sent)
        assert it_a == it_v, (n, it_a, it_v)
        rows.append(it_a)
    meta["sched"] = rows


def _ids(world, toks):
    return [int(world.tok.convert_tokens_to_ids(t)) for t in toks]


def word_info(world, text):
    """Id-level view of a sentence the way ``_tokenize`` (adv_attack.py:141-154) splits it: per whitespace word its
    word-piece ids, its [start, end) span in piece coordinates and whether the word string is a filter word."""
    words = text.replace("\n", "").lower().split(" ")
    pieces, keys, at = [], [], 0
    for w in words:
        ids = _ids(world, world.tok.tokenize(w))
        pieces.append(ids)
        keys.append([at, at + len(ids)])
        at += len(ids)
    fw = set(world.filter_words)
    return dict(words=words, word_ids=pieces, keys=keys, word_filtered=[w in fw for w in words])


CAND_SENTENCES = ["what color is the cat", "is the man holding a red umbrella near the table",
                  "how many cats are playing on the table", "the the the", "zebras playing frisbee",
                  "what is the woman eating in the kitchen", "red table cat man umbrella playing"]


def section_candidates(world, arrays, meta):
    """cal_text_attack_list (iter_list + candidate words) per sentence, both copies, with everything the id-level
    restatement needs: the ids / logits the MLM stand-in saw, word -> piece spans."""
    sa, _ = albef_stub(world)
    sv, _ = vlmo_stub(world)
    cases = []
    for flavor, stub, suffix in (("albef", sa, ""), ("vlmo", sv, "?")):
        for s in CAND_SENTENCES:
            text = s + suffix
            iters, cands = stub.cal_text_attack_list(text)
            words, sub_words, keys = stub._tokenize(text.strip("?").lower() if flavor == "vlmo" else text.lower(),
                                                    world.tok)
            ids = _ids(world, ["[CLS]"] + sub_words + ["[SEP]"])
            info = word_info(world, text.strip("?") if flavor == "vlmo" else text)
            assert info["keys"] == keys and info["words"] == words
            cases.append(dict(flavor=flavor, text=text, words=words, sub_words=sub_words, keys=keys, mlm_input_ids=ids,
                              word_filtered=info["word_filtered"], iter_list=iters, cand_words=cands,
                              cand_ids=[None if c is None else _ids(world, c) for c in cands]))
    meta["cand_cases"] = cases
    # get_substitues on its own: empty, single-piece with the threshold cut, multi-piece (BPE branch through the MLM)
    subs = torch.tensor([[5 + 104, 7 + 104, 9 + 104, 11 + 104, 13 + 104]])
    scores = torch.tensor([[0.9, 0.5, 0.31, 0.29, 0.8]])
    meta["getsub_single"] = dict(ids=subs.tolist(), scores=scores.tolist(),
                                 words=sa.get_substitues(subs, world.tok, world.mlm, substitutes_score=scores))
    meta["getsub_empty"] = sa.get_substitues(torch.zeros(0, 5, dtype=torch.long), world.tok, world.mlm,
                                             substitutes_score=torch.zeros(0, 5))
    multi = torch.tensor([[110, 111, 112, 113, 114], [120, 121, 122, 123, 124]])
    meta["getsub_multi"] = dict(ids=multi.tolist(),
                                words=sa.get_substitues(multi, world.tok, world.mlm,
                                                        substitutes_score=torch.ones(2, 5)))


def _question_setup(world, stub, flavor, question):
    """What evaluate / test_step compute before the loop: candidate lists, attack_vector, sub_list, ids, ori_emb."""
    iters, cands = stub.cal_text_attack_list(question)
    words, _, keys = stub._tokenize(question.strip("?").lower() if flavor == "vlmo" else question.lower(), world.tok)
    attack_vector, sub_list = [], []
    for idx, (key, cand) in enumerate(zip(keys, cands)):
        if cand is not None:
            attack_vector.append(key[0] + 1)
            sub_list.append(idx)
    if flavor == "albef":
        enc = world.tok(question, padding="longest", truncation=True, max_length=25, return_tensors="pt")
        ids = enc["input_ids"]
    else:
        enc = world.tok(question, padding="max_length", truncation=True, max_length=40, return_special_tokens_mask=True)
        ids = torch.tensor(enc["input_ids"]).unsqueeze(0)
    return iters, cands, attack_vector, sub_list, ids


def section_update(world, arrays, meta):
    """update_adv_text over several questions and synthetic text gradients, both copies; update_mlm_text."""
    g = np.random.RandomState(9)
    cases = []
    qs = ["what color is the cat", "is the man holding a red umbrella near the table",
          "red table cat man umbrella playing"]
    for flavor, mk, suffix in (("albef", albef_stub, ""), ("vlmo", vlmo_stub, "?")):
        stub, _ = mk(world, 64)
        for qi, q in enumerate(qs):
            question = q + suffix
            iters, cands, attack_vector, sub_list, ids = _question_setup(world, stub, flavor, question)
            with torch.no_grad(), rx.cpu_as_cuda():
                ori_emb = stub.text_embeddings(ids)
                adv_text = question
                rounds = []
                for rnd in range(2):                     # two consecutive rounds: the second starts from the adv text
                    grad = g.standard_normal((1, len(attack_vector), 64)).astype(np.float32)
                    new_text, ops = stub.update_adv_text(torch.from_numpy(grad), cands, sub_list, adv_text,
                                                         attack_vector, ori_emb, question)
                    if flavor == "albef":
                        new_ids = world.tok(new_text, padding="longest", truncation=True, max_length=25)["input_ids"]
                        cur_ids = world.tok(adv_text, padding="longest", truncation=True, max_length=25)["input_ids"]
                    else:
                        new_ids = world.tok(new_text, padding="max_length", truncation=True, max_length=40)["input_ids"]
                        cur_ids = world.tok(adv_text, padding="max_length", truncation=True, max_length=40)["input_ids"]
                    key = "upd_{}_{}_{}".format(flavor, qi, rnd)
                    arrays[key + "_grad"] = grad
                    rounds.append(dict(key=key, adv_text_in=adv_text, cur_ids=cur_ids, adv_text_out=new_text,
                                       new_ids=new_ids, op_mlm_list=ops, op_ids=[_ids(world, o) for o in ops]))
                    adv_text = new_text
            cases.append(dict(flavor=flavor, question=question, ori_ids=ids[0].tolist(), attack_vector=attack_vector,
                              sub_list=sub_list, cand_ids=[None if c is None else _ids(world, c) for c in cands],
                              cand_words=cands, rounds=rounds))
    meta["upd_cases"] = cases
    # update_mlm_text: word replacement in the [MASK]-ed paraphrase + re-encoding
    out = []
    for flavor, mk in (("albef", albef_stub), ("vlmo", vlmo_stub)):
        stub, _ = mk(world, 64)
        list_words = "the cat is [MASK] on the red table near the cat".split()
        ops = [["cat", "dog"], ["red", "blue"], ["zebra", "horse"]]
        with rx.cpu_as_cuda():
            stub.update_mlm_text(ops, list_words)
        out.append(dict(flavor=flavor, ops=ops, list_words_out=list_words,
                        text_ids_mlm=stub.batch["text_ids_mlm"].tolist(),
                        text_mask_mlm=stub.batch["text_mask_mlm"].tolist()))
    meta["updmlm_cases"] = out


def main(out_npz=None, out_json=None):
    out_npz, out_json = out_npz or OUT_NPZ, out_json or OUT_JSON
    torch.manual_seed(0)
    world = World()
    arrays, meta = {}, {}
    meta["vocab"] = world.vocab
    meta["filter_in_vocab"] = world.filter_in_vocab
    arrays["use_table"] = world.use_table
    arrays["mlm_table"] = world.mlm.table
    arrays["mlm_drift"] = world.mlm.drift
    section_embeddings(world, arrays, meta)
    section_dir_sim(world, arrays, meta)
    section_schedule(world, arrays, meta)
    section_candidates(world, arrays, meta)
    section_update(world, arrays, meta)
    from tests.golden import make_text_golden_tasks as tasks
    tasks.run(world, arrays, meta)
    np.savez_compressed(out_npz, **arrays)
    with open(out_json, "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("wrote", out_npz, "({} arrays, {:.1f} KB)".format(len(arrays), os.path.getsize(out_npz) / 1024))
    print("wrote", out_json, "({:.1f} KB)".format(os.path.getsize(out_json) / 1024))


if __name__ == "__main__":
    main()
