"""Host-side text logic of the SHIPPED path (no HIP kernel involved, runs without a GPU) against the vectors produced by
executing the reference's own methods (``tests/golden/text_golden.*``): candidate proposal, block schedule, loss-mode
decision + MLM task construction, MLM-text update.  The oracle is not involved here."""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden import textworld as tw
from vqattack_amd.attack import mlm_task, schedule, text_update

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(HERE, "golden", "text_golden.npz"))
    with open(os.path.join(HERE, "golden", "text_golden.json")) as fh:
        meta = json.load(fh)
    return z, meta


def banned_mask(meta):
    fw = set(meta["filter_in_vocab"])
    return torch.tensor([("##" in tok) or (tok in fw) for tok in meta["vocab"]])


def test_schedule_equals_reference(gold):
    _, meta = gold
    for n, want in enumerate(meta["sched"]):
        assert schedule.iter_schedule(n) == want
        assert schedule.gradient_steps(n) == 40 + n


def test_propose_candidates_equals_cal_text_attack_list(gold):
    z, meta = gold
    mlm = tw.MlmStandIn(z["mlm_table"], z["mlm_drift"])
    banned = banned_mask(meta)
    n_filtered_slots = 0
    for c in meta["cand_cases"]:
        ids = c["mlm_input_ids"]
        subs = text_update.substitutable_words(c["keys"], c["word_filtered"])
        assert schedule.iter_schedule(len(subs)) == c["iter_list"], c["text"]
        attackable = torch.zeros(1, len(ids), dtype=torch.bool)
        for w in subs:
            attackable[0, c["keys"][w][0] + 1] = True
        logits = torch.from_numpy(mlm.logits_for(ids)).unsqueeze(0)
        got = text_update.propose_candidates(logits, torch.tensor([ids]), attackable, banned=banned)[0]
        want = [(k[0] + 1, c["cand_ids"][i]) for i, k in enumerate(c["keys"])
                if i < len(c["cand_ids"]) and c["cand_ids"][i] is not None]
        assert got == want, c["text"]
        n_filtered_slots += sum(5 - len(v) for _, v in want)
    assert n_filtered_slots > 0      # threshold cut / original word / piece / stop word filters were exercised


def _words(ws):
    return [tuple(w) for w in ws]


def test_build_mlm_task_equals_reference(gold):
    _, meta = gold
    n_dual = n_3d = 0
    for t in meta["task_cases"]:
        flavor = t["flavor"]
        task = mlm_task.build_mlm_task(_words(t["ans_word_ids"]), [_words(a) for a in t["all_ans_word_ids"]],
                                       t["same_as_vilt"], _words(t["pa_word_ids"]), _words(t["stop_word_ids"]), flavor,
                                       tail=() if flavor == "albef" else (t["period_id"],))
        assert task.old_alg == t["old_alg"], (flavor, t["paraphrase"])
        if task.old_alg == 1:
            assert task.mlm_labels is None
            continue
        n_dual += 1
        assert [task.text_ids_mlm] == t["text_ids_mlm"] and [task.text_mask_mlm] == t["text_mask_mlm"]
        want = t["mlm_labels"][0]
        assert task.mlm_labels == want
        n_3d += isinstance(want[0], list)
        vocab = meta["vocab"]
        assert [" ".join(vocab[i] for i in w).replace(" ##", "") for w in task.words_mlm] == t["list_words"]
    assert n_dual >= 10 and n_3d >= 6


def test_empty_answer_set_is_an_error_like_the_reference():
    with pytest.raises(UnboundLocalError):
        mlm_task.build_mlm_task([(7,)], [], [], [(5,), (7,)], [], "albef")


def test_apply_substitutions_equals_update_mlm_text(gold):
    _, meta = gold
    vocab = {t: i for i, t in enumerate(meta["vocab"])}
    for c in meta["updmlm_cases"]:
        words = [(vocab[w],) for w in "the cat is [MASK] on the red table near the cat".split()]
        words[1] = (vocab["cat"],)
        out = mlm_task.apply_substitutions(words, [(vocab[a], vocab[b]) for a, b in c["ops"]])
        assert out == [(vocab[w],) for w in c["list_words_out"]]
        ids, mask = mlm_task.encode(out, c["flavor"], tail=() if c["flavor"] == "albef" else (vocab["."],))
        assert [ids] == c["text_ids_mlm"] and [mask] == c["text_mask_mlm"]
    # a one-piece word is not replaced inside a multi-piece word ("cat" vs "cat ##s")
    words = [(vocab["cat"], vocab["##s"]), (vocab["cat"],)]
    assert mlm_task.apply_substitutions(words, [(vocab["cat"], vocab["dog"])]) == [(vocab["cat"], vocab["##s"]),
                                                                                   (vocab["dog"],)]
