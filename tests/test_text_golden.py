"""The text-side CPU oracle against vectors produced by EXECUTING the reference's own methods
(``tests/golden/make_text_golden.py``: ast-compiled from the reference source, run in the build container).

Every comparison here is oracle (``oracle/text_scoring.py``, ``oracle/attack_loop.py``, ``oracle/adapters_ref.py``,
``oracle/blackbox_ref.py``) vs fixture; the HIP path is compared with the same fixtures in ``tests/test_text_hip.py``.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import text_scoring as ts
from tests.golden import textworld as tw

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(HERE, "golden", "text_golden.npz"))
    with open(os.path.join(HERE, "golden", "text_golden.json")) as fh:
        meta = json.load(fh)
    return z, meta


def tables_of(z, dim):
    t = {k: torch.from_numpy(z["tab{}_{}".format(dim, k)]) for k in ("word", "pos", "type_emb", "gamma", "beta")}
    return (t["word"], t["pos"], t["type_emb"], t["gamma"], t["beta"], 1e-12)


def banned_tokens(meta):
    fw = set(meta["filter_in_vocab"])
    return [("##" in tok) or (tok in fw) for tok in meta["vocab"]]


def mlm_logits(z, ids):
    return tw.MlmStandIn(z["mlm_table"], z["mlm_drift"]).logits_for(ids)


def similarity(z):
    enc = tw.SentenceEncoderStandIn(None, z["use_table"])
    return enc.similarity_ids


# ----------------------------------------------------------------------------------------- embeddings / dir_sim
@pytest.mark.parametrize("dim", [64, 768])
def test_bert_embeddings_equal_reference(gold, dim):
    z, _ = gold
    out = ts.bert_embeddings(torch.from_numpy(z["emb_ids_{}".format(dim)]), *tables_of(z, dim))
    assert torch.equal(out, torch.from_numpy(z["emb_out_{}".format(dim)]))       # bitwise: same ATen ops, same order


def test_dir_sim_equals_reference(gold):
    z, _ = gold
    a, b = torch.from_numpy(z["dirsim_a"]), torch.from_numpy(z["dirsim_b"])
    got = torch.stack([ts.dir_sim(x, y) for x, y in zip(a, b)])
    assert torch.equal(got, torch.from_numpy(z["dirsim_out"]))


# ----------------------------------------------------------------------------------------- schedule / candidates
def test_iter_schedule_for_0_to_40_words(gold):
    _, meta = gold
    for n, want in enumerate(meta["sched"]):
        assert ts.iter_schedule(n) == want, n
        if want:
            assert sum(want) == 40 and len(want) == n + 1


def test_cal_text_attack_list_equals_reference(gold):
    z, meta = gold
    banned = banned_tokens(meta)
    assert len(meta["cand_cases"]) >= 14
    for c in meta["cand_cases"]:
        ids = c["mlm_input_ids"]
        iters, cand = ts.cal_text_attack_list(ids, mlm_logits(z, ids), c["keys"], c["word_filtered"], banned)
        assert iters == c["iter_list"], c["text"]
        assert cand == c["cand_ids"], c["text"]


def test_get_substitutes_threshold_cut(gold):
    _, meta = gold
    g = meta["getsub_single"]
    vocab = meta["vocab"]
    got = ts.get_substitutes(g["ids"], g["scores"])
    assert [vocab[i] for i in got] == g["words"] and len(got) == 3       # cut at the first score < 0.3, later 0.8 unseen
    assert ts.get_substitutes([], []) == meta["getsub_empty"] == []


# ----------------------------------------------------------------------------------------- update_adv_text
def test_update_adv_text_equals_reference(gold):
    z, meta = gold
    tabs = tables_of(z, 64)
    sim = similarity(z)
    n_ops = 0
    for c in meta["upd_cases"]:
        ori = c["ori_ids"]
        e_ori = ts.bert_embeddings(torch.tensor([ori]), *tabs)
        for r in c["rounds"]:
            grad = torch.from_numpy(z[r["key"] + "_grad"])
            new_ids, ops = ts.update_adv_text(grad, c["cand_ids"], c["sub_list"], c["attack_vector"], r["cur_ids"], ori,
                                              e_ori, tabs, sim)
            assert new_ids == r["new_ids"], (c["flavor"], c["question"], r["adv_text_in"])
            assert [list(o) for o in ops] == r["op_ids"]
            n_ops += len(ops)
    assert n_ops >= 4          # the fixtures do exercise accepted substitutions


def test_update_mlm_text_equals_reference(gold):
    _, meta = gold
    vocab = {t: i for i, t in enumerate(meta["vocab"])}
    for c in meta["updmlm_cases"]:
        words = [(vocab[w],) for w in "the cat is [MASK] on the red table near the cat".split()]
        ops = [(vocab[a], vocab[b]) for a, b in c["ops"]]
        out = ts.update_mlm_text(ops, words)
        assert out == [(vocab[w],) for w in c["list_words_out"]]
        flavor = c["flavor"]
        ids, mask = ts.encode_words(out, 25 if flavor == "albef" else 40, None if flavor == "albef" else 40,
                                    tail=() if flavor == "albef" else (vocab["."],))
        assert [ids] == c["text_ids_mlm"] and [mask] == c["text_mask_mlm"]


# ----------------------------------------------------------------------------------------- MLM task construction
def _task_args(t):
    tup = lambda ws: [tuple(w) for w in ws]      # noqa: E731
    return dict(ans_words=tup(t["ans_word_ids"]), all_ans_words=[tup(a) for a in t["all_ans_word_ids"]],
                same_as_vilt=t["same_as_vilt"], pa_words=tup(t["pa_word_ids"]), stop_words=tup(t["stop_word_ids"]))


def test_build_mlm_task_equals_reference(gold):
    _, meta = gold
    seen = {0: 0, 1: 0}
    three_d = 0
    for t in meta["task_cases"]:
        assert t["error"] is None
        flavor = t["flavor"]
        got = ts.build_mlm_task(flavor=flavor, tail=() if flavor == "albef" else (t["period_id"],), **_task_args(t))
        assert got["old_alg"] == t["old_alg"], (flavor, t["paraphrase"])
        seen[t["old_alg"]] += 1
        if t["old_alg"] == 1:
            continue
        assert [got["text_ids_mlm"]] == t["text_ids_mlm"]
        assert [got["text_mask_mlm"]] == t["text_mask_mlm"]
        want = t["mlm_labels"]                                  # (1, L) or (1, K, L)
        if isinstance(want[0][0], list):
            assert got["mlm_labels"] == want[0]
            three_d += 1
        else:
            assert got["mlm_labels"] == want[0] or got["mlm_labels"] == [want[0]]
        vocab = meta["vocab"]
        assert [" ".join(vocab[i] for i in w).replace(" ##", "") for w in got["list_words"]] == t["list_words"]
    assert seen[0] >= 10 and seen[1] >= 2 and three_d >= 6
