"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the text side of the joint attack (rows a10/a11 of SURVEY.md 8a).

The reference works on strings (tokenizer round trips); this restatement works on token ids, which is what the
shipped path sees.  The two agree because the reference only ever substitutes a word that is ONE word piece by a
candidate that is ONE vocabulary token (``adv_attack.py:226,252``), so a substitution never moves another token.

**Pinned** by ``tests/golden/text_golden.{npz,json}``: vectors produced by EXECUTING the reference's own methods (compiled
from the reference source with ``ast``, ``tests/golden/make_text_golden.py``) over a synthetic text world --
``tests/test_text_golden.py`` requires equality.  Assumptions under which the id-level restatement equals the string
level (all hold for the fixtures; the first two are properties of BERT WordPiece):
  * tokenising a whitespace-joined sentence = concatenating the tokenisations of its words;
  * a candidate token re-tokenises to itself (no punctuation inside a candidate);
  * two different words do not both map to [UNK].

  * iter_schedule ........... ALBEF_attack/adv_attack.py:229-239 (= vlmo_module.py:1545-1556)
  * bert_embeddings ......... ALBEF_attack/models/xbert.py:189-216 (BertEmbeddings, eval mode -> no dropout)
  * dir_sim ................. ALBEF_attack/adv_attack.py:325-333 (= vlmo_module.py:1632-1640)
  * get_substitutes ......... ALBEF_attack/adv_attack.py:191-207 (single-piece branch; the BPE branch is unreachable
                              from cal_text_attack_list, whose substitute_list only holds single-piece words)
  * cal_text_attack_list .... ALBEF_attack/adv_attack.py:215-264 (= vlmo_module.py:1531-1581)
  * candidate_scores ........ ALBEF_attack/adv_attack.py:272-298 for position-preserving single-token substitutions
  * update_adv_text ......... ALBEF_attack/adv_attack.py:265-324 (= vlmo_module.py:1642-1702)
  * update_mlm_text ......... ALBEF_attack/adv_attack.py:334-351 (= vlmo_module.py:1704-1723)
  * build_mlm_task .......... ALBEF_attack/adv_attack.py:433-558 (= vlmo_module.py:1748-1889)
"""
import copy

import torch
import torch.nn.functional as F

PAD, CLS, SEP, MASK = 0, 101, 102, 103
ANSWER_STOP_WORDS = ("on", "and", "in", "his", "her", "its")     # Adv_attack.filter, adv_attack.py:155-160


def iter_schedule(n_attackable_words, budget=40):
    """PGD steps per block for a question with ``n_attackable_words`` substitutable words (0 -> no blocks)."""
    if n_attackable_words == 0:
        return []
    count = n_attackable_words + 1
    q = int(budget / count)
    steps = [q if q % 2 == 0 else q - 1 for _ in range(count)]
    steps[-1] += budget - sum(steps)
    return steps


def bert_embeddings(ids, word, pos, type_emb, gamma, beta, ln_eps):
    """LayerNorm(word[ids] + type[0] + pos[0..L)) for ids (B, L)."""
    length = ids.shape[1]
    e = F.embedding(ids, word) + type_emb[0]
    e = e + pos[:length].unsqueeze(0)
    return F.layer_norm(e, (word.shape[1],), gamma, beta, ln_eps)


def dir_sim(cand_emb_dir, attack_grad):
    a = F.normalize(cand_emb_dir, p=2, dim=0)
    b = F.normalize(attack_grad, p=2, dim=0)
    return torch.nn.CosineSimilarity(dim=0, eps=1e-6)(a, b)


def get_substitutes(sub_ids, sub_scores, threshold=0.3):
    """adv_attack.py:191-207 for a one-piece word: top-k ids in order, cut at the first score below the threshold."""
    out = []
    if len(sub_ids) != 1:
        return out            # 0 pieces: nothing; > 1 piece: the BPE branch, never reached from cal_text_attack_list
    for i, j in zip(sub_ids[0], sub_scores[0]):
        if threshold != 0 and j < threshold:
            break
        out.append(int(i))
    return out


def cal_text_attack_list(input_ids, logits, keys, word_is_filtered, banned_token, budget=40, topk=5):
    """``(iter_list, candidate ids per word or None)``.

    ``input_ids``: [CLS] + word pieces + [SEP] of the question; ``logits`` (len(input_ids), V): the MLM's output for
    them; ``keys``: per word its ``[start, end)`` span in word-piece coordinates (``_tokenize``); ``word_is_filtered``:
    per word, whether the WORD STRING is in filter_words; ``banned_token`` (V,) bool: vocabulary entries that are a
    ``##`` piece or a filter word (never proposed)."""
    n_sub = len(input_ids) - 2
    cand = [None for _ in keys]
    substitute_list = [k for k, f in zip(keys, word_is_filtered) if k[1] - k[0] == 1 and not f]
    count = len(substitute_list)
    if count == 0:
        return [], []
    iters = iter_schedule(count, budget)
    scores_all, preds = torch.topk(torch.as_tensor(logits), topk, -1)
    preds = preds[1:n_sub + 1 + 1, :]               # word_predictions[1:len(sub_words) + 1] with sub_words incl. CLS/SEP
    scores_all = scores_all[1:n_sub + 1 + 1, :]
    for sub in substitute_list:
        w = keys.index(sub)
        original = int(input_ids[1 + sub[0]])
        for v in get_substitutes(preds[sub[0]:sub[1]].tolist(), scores_all[sub[0]:sub[1]].tolist()):
            if v == original:
                continue                              # filter out original word
            if bool(banned_token[v]):
                continue                              # '##' piece or filter word
            if cand[w] is None:
                cand[w] = []
            cand[w].append(v)
    return iters, cand


def candidate_scores(ids, e_ori, grad, cand, word, pos, type_emb, gamma, beta, ln_eps):
    """cand: int (n, 4) rows {sample, position, grad row, vocabulary id}; returns fp32 (n,).
    For each candidate the sentence is re-embedded with the candidate id written at ``position`` and the embedding
    direction at that position is compared with the text gradient row (adv_attack.py:284-298)."""
    out = []
    for s, p, k, v in cand.tolist():
        sent = ids[s:s + 1].clone()
        sent[0, p] = v
        e = bert_embeddings(sent, word, pos, type_emb, gamma, beta, ln_eps)
        out.append(dir_sim(e[0, p] - e_ori[s, p], grad[s, k]))
    return torch.stack(out) if out else torch.zeros(0)


def update_adv_text(text_grad, cand_ids, sub_list, attack_vector, cur_ids, ori_ids, e_ori, tables, similarity_fn,
                    sim_threshold=0.95):
    """One substitution round for one sample; returns ``(new ids list, op list [(old id, new id), ...])``.

    ``text_grad`` (1, K, D): row idx belongs to ``attack_vector[idx]``; ``cand_ids``: per word a list or None;
    ``sub_list``: word indices that have candidates; ``attack_vector``: their token positions; ``cur_ids`` / ``ori_ids``:
    id lists of the current adversarial / the original question; ``e_ori`` (1, L, D) embeddings of the original;
    ``similarity_fn(ori_ids, trial_ids) -> float`` stands for the sentence encoder."""
    word, pos, type_emb, gamma, beta, ln_eps = tables
    cur = torch.tensor([list(cur_ids)], dtype=torch.long)
    sort_list, sims = [], []
    for idx, (w, p) in enumerate(zip(sub_list, attack_vector)):
        g = text_grad[0, idx]
        for idd, v in enumerate(cand_ids[w]):
            sort_list.append((w, idd, p))
            sent = cur.clone()
            sent[0, p] = v
            e = bert_embeddings(sent, word, pos, type_emb, gamma, beta, ln_eps)
            sims.append(dir_sim(e[0, p] - e_ori[0, p], g))
    order = sorted(range(len(sims)), key=lambda k: sims[k], reverse=True)
    new_ids = list(cur_ids)
    before = list(cur_ids)
    occupied, ops = [], []
    thr = sim_threshold
    for k in order:
        w, idd, p = sort_list[k]
        if w in occupied:
            continue
        trial = list(new_ids)
        trial[p] = cand_ids[w][idd]
        sim = similarity_fn([int(t) for t in ori_ids], trial)
        if sim > thr:
            thr = sim
            occupied.append(w)
            new_ids = trial
            ops.append((int(before[p]), int(cand_ids[w][idd])))
    return new_ids, ops


def update_mlm_text(ops, list_words):
    """adv_attack.py:334-341 on a list of words given as tuples of piece ids: every occurrence of a replaced word
    (a one-piece word) is replaced in the [MASK]-ed paraphrase.  Mutates and returns ``list_words``."""
    for old, new in ops:
        if (old,) in list_words:
            for i, w in enumerate(list_words):
                if w == (old,):
                    list_words[i] = (new,)
    return list_words


def encode_words(words, max_len, pad_to=None, tail=()):
    """tokenizer(' '.join(words) [+ '.']) on id tuples: [CLS] pieces... tail [SEP], truncated to ``max_len``, padded to
    ``pad_to`` (None = 'longest', i.e. no padding)."""
    body = [t for w in words for t in w] + list(tail)
    body = body[:max_len - 2]
    ids = [CLS] + body + [SEP]
    mask = [1] * len(ids)
    if pad_to is not None:
        ids += [PAD] * (pad_to - len(ids))
        mask += [0] * (pad_to - len(mask))
    return ids, mask


def _filter(words, stop_words):
    """Adv_attack.filter: removes the FIRST occurrence of each stop word, in place."""
    for s in stop_words:
        if s in words:
            words.remove(s)
    return words


def build_mlm_task(ans_words, all_ans_words, same_as_vilt, pa_words, stop_words, flavor, tail=()):
    """old_alg decision + [MASK]-ed paraphrase + MLM labels (adv_attack.py:433-558; vlmo_module.py:1748-1889).

    Words are tuples of word-piece ids.  ``ans_words``: the victim answer's words; ``all_ans_words``: one word list per
    entry of all_correct_ans; ``same_as_vilt``: per entry whether its STRING equals the victim answer; ``pa_words``: the
    paraphrase's words; ``stop_words``: the six answer stop words as id tuples; ``tail``: pieces appended before [SEP]
    (VLMO appends '.').  Returns dict(old_alg, text_ids_mlm, text_mask_mlm, mlm_labels (L,) or (K, L) or None,
    list_words, mask_pos, sub_words_length)."""
    max_len, pad_to = (25, None) if flavor == "albef" else (40, 40)

    def keys_of(words):
        out, at = [], 0
        for w in words:
            out.append([at, at + len(w)])
            at += len(w)
        return out

    pa_words = list(pa_words)
    pa_keys = keys_of(pa_words)
    attack_ans_words = _filter(list(ans_words), stop_words)
    encode_ids, _ = encode_words(pa_words, max_len, pad_to, tail)
    label_len = len(encode_ids) if flavor == "albef" else 40
    labels = [-100] * label_len
    mask_pos_list, mask_word_list, sub_len_lst, vilt_ans_word_lst = [], [], [], []
    old_alg, mask_pos, sub_words_length = 1, None, None
    for a in attack_ans_words:
        if a in pa_words:
            vilt_ans_word_lst.append(a)
            old_alg = 0
            mask_pos = pa_words.index(a)
            mask_pos_list.append(mask_pos)
            sub_words_length = pa_keys[mask_pos][-1] - pa_keys[mask_pos][0]
            sub_len_lst.append(sub_words_length)
            mask_word_list.append([(MASK,)] * sub_words_length)
    vilt_pos_list = list(mask_pos_list)
    if old_alg == 1:
        return dict(old_alg=1, mlm_labels=None, text_ids_mlm=None, text_mask_mlm=None, list_words=None,
                    mask_pos=None, sub_words_length=None)

    def write_labels(dst, keys, mp, src):
        lo, hi = keys[mp][0] + 1, keys[mp][1] + 1
        seg = src[lo:hi]
        dst[lo:lo + len(seg)] = seg          # slice assignment of equal length (tensor semantics)

    list_words = list(pa_words)
    order = sorted(range(len(mask_pos_list)), key=lambda k: mask_pos_list[k], reverse=True)
    mask_pos_list = sorted(mask_pos_list, reverse=True)
    new_mask_word_list = [mask_word_list[i] for i in order]
    for mp, sub in zip(mask_pos_list, new_mask_word_list):
        list_words = list_words[0:mp] + sub + list_words[mp + 1:]
        write_labels(labels, pa_keys, mp, encode_ids)
    ids_mlm, mask_mlm = encode_words(list_words, max_len, pad_to, tail)
    if len(all_ans_words) == 1:
        mlm_labels = labels
    elif len(all_ans_words) > 1:
        sets = [labels]
        for cand_words, same in zip(all_ans_words, same_as_vilt):
            cand_words = list(cand_words)
            cand_keys = keys_of(cand_words)                     # keys of the UNFILTERED answer (reference quirk)
            cand_attack = _filter(cand_words, stop_words)
            if len(cand_attack) != len(vilt_ans_word_lst):
                continue
            if same:
                continue
            ok, cand_pos, cand_masks = True, [], []
            for i, _w in enumerate(cand_attack):
                n = cand_keys[i][-1] - cand_keys[i][0]
                if n != sub_len_lst[i]:
                    ok = False
                    break
                cand_masks.append([(MASK,)] * n)
                cand_pos.append(vilt_pos_list[i])
            if not ok:
                continue
            cand_pos = sorted(cand_pos, reverse=True)
            cand_labels = [-100] * label_len
            cand_pa_words = list(pa_words)
            cand_pa_keys = keys_of(cand_pa_words)
            for cli, cwl in zip(cand_pos, cand_attack):        # descending positions zipped with words in answer order
                cand_pa_words[cli] = cwl
            cand_encode, _ = encode_words(cand_pa_words, max_len, pad_to, tail)
            for mp in cand_pos:
                write_labels(cand_labels, cand_pa_keys, mp, cand_encode)
            sets.append(cand_labels)
        mlm_labels = sets[0] if len(sets) == 1 else sets
    else:
        raise UnboundLocalError("mlm_labels is unbound for an empty all_correct_ans (as in the reference)")
    return dict(old_alg=0, mlm_labels=mlm_labels, text_ids_mlm=ids_mlm, text_mask_mlm=mask_mlm, list_words=list_words,
                mask_pos=mask_pos, sub_words_length=sub_words_length)
