"""Loss-mode decision and MLM task of a sample, on token ids (row a11 of SURVEY.md section 8a).

Reference: ``Adv_attack.evaluate`` (``ALBEF_attack/adv_attack.py:433-558``) / ``VLMo.test_step``
(``vlmo/modules/vlmo_module.py:1748-1889``).  For every question the reference looks up the victim model's answer and a
declarative paraphrase of the question; if a (non-stop) word of the answer occurs in the paraphrase the sample is
attacked with the dual loss (``old_alg = 0``): the answer word's pieces are replaced by [MASK] in the paraphrase, the
MLM labels are the masked pieces, and every other correct answer whose words have the same piece counts contributes one
more label set (3-d labels).  Otherwise (``old_alg = 1``) only the feature loss is used.

The reference does this on strings with tokenizer round trips; here a word is the tuple of its word-piece ids (what the
caller's tokenizer gives per whitespace-separated word), which is all the arithmetic needs.  The reference's quirks are
kept (each is exercised by the reference-generated fixtures in ``tests/golden/text_golden.json``):
  * ``filter`` drops only the FIRST occurrence of each of its six stop words;
  * an alternative answer's piece counts are read from the spans of its UNFILTERED word list;
  * alternative words are written at the mask positions sorted in DESCENDING order but taken in answer order;
  * an empty answer set leaves the labels undefined (``UnboundLocalError``).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

PAD_ID, CLS_ID, SEP_ID, MASK_ID = 0, 101, 102, 103
IGNORE = -100
Word = Tuple[int, ...]


@dataclass
class MlmTask:
    old_alg: int                               # 1 = feature loss only, 0 = feature + MLM dual loss
    text_ids_mlm: Optional[List[int]] = None   # [MASK]-ed paraphrase, encoded ([CLS] ... [SEP] + padding)
    text_mask_mlm: Optional[List[int]] = None
    mlm_labels: Optional[list] = None          # (L,) or (K, L) nested lists
    words_mlm: Optional[List[Word]] = None     # the paraphrase's word list with [MASK] words (update_mlm_text edits it)
    mask_pos: Optional[int] = None             # word index of the last masked answer word
    n_masked_pieces: Optional[int] = None
    flavor: str = "albef"                      # how the paraphrase is (re-)encoded: see encode()
    tail: Tuple[int, ...] = ()
    max_len: Optional[int] = None

    def reencode(self):
        """ids / mask of the current ``words_mlm`` (after ``apply_substitutions``)."""
        self.text_ids_mlm, self.text_mask_mlm = encode(self.words_mlm, self.flavor, self.tail, self.max_len)
        return self.text_ids_mlm


def _spans(words: Sequence[Word]):
    spans, at = [], 0
    for w in words:
        spans.append((at, at + len(w)))
        at += len(w)
    return spans


def drop_answer_stop_words(words: List[Word], stop_words: Sequence[Word]) -> List[Word]:
    """``Adv_attack.filter`` (adv_attack.py:155-160): first occurrence of each stop word, in place."""
    for s in stop_words:
        if s in words:
            words.remove(s)
    return words


def encode(words: Sequence[Word], flavor: str, tail: Sequence[int] = (), max_len: Optional[int] = None):
    """ids / mask of ``' '.join(words)`` (+ the VLMO copy's trailing '.') as the reference's tokenizer calls give them:
    ALBEF ``padding='longest', max_length=25``; VLMO ``padding='max_length', max_length=40`` (``max_len`` overrides the
    literal for white boxes built with another text length)."""
    max_len = max_len or (25 if flavor == "albef" else 40)
    body = ([t for w in words for t in w] + list(tail))[:max_len - 2]
    ids = [CLS_ID] + body + [SEP_ID]
    mask = [1] * len(ids)
    if flavor != "albef":
        pad = max_len - len(ids)
        ids, mask = ids + [PAD_ID] * pad, mask + [0] * pad
    return ids, mask


def _label_row(length, spans, positions, encoded):
    row = [IGNORE] * length
    for wpos in positions:
        lo, hi = spans[wpos][0] + 1, spans[wpos][1] + 1       # + 1: [CLS]
        seg = encoded[lo:hi]
        row[lo:lo + len(seg)] = seg
    return row


def build_mlm_task(answer_words: Sequence[Word], correct_answers: Sequence[Sequence[Word]],
                   is_victim_answer: Sequence[bool], paraphrase_words: Sequence[Word], stop_words: Sequence[Word],
                   flavor: str, tail: Sequence[int] = (), max_len: Optional[int] = None) -> MlmTask:
    """``answer_words``: words of the victim's answer; ``correct_answers``: words of every entry of the sample's
    correct-answer list, ``is_victim_answer[i]`` telling whether entry i is literally the victim's answer;
    ``paraphrase_words``: words of the paraphrase (trailing '.' stripped); ``tail``: piece ids appended before [SEP]."""
    para = list(paraphrase_words)
    spans = _spans(para)
    wanted = drop_answer_stop_words(list(answer_words), stop_words)
    full_ids, _ = encode(para, flavor, tail, max_len)
    n_labels = len(full_ids)
    hit_pos, hit_len, hit_words = [], [], []
    for w in wanted:
        if w in para:
            hit_words.append(w)
            hit_pos.append(para.index(w))
            hit_len.append(spans[hit_pos[-1]][1] - spans[hit_pos[-1]][0])
    if not hit_pos:
        return MlmTask(old_alg=1)
    # replace the answer words by [MASK] pieces, right to left so that earlier word indices stay valid
    masked = list(para)
    for wpos, n in sorted(zip(hit_pos, hit_len), key=lambda t: t[0], reverse=True):
        masked = masked[:wpos] + [(MASK_ID,)] * n + masked[wpos + 1:]
    ids_mlm, mask_mlm = encode(masked, flavor, tail, max_len)
    base = _label_row(n_labels, spans, sorted(hit_pos, reverse=True), full_ids)
    if len(correct_answers) == 0:
        raise UnboundLocalError("mlm_labels is undefined for an empty correct-answer list (as in the reference)")
    label_sets = [base]
    if len(correct_answers) > 1:
        for alt, same in zip(correct_answers, is_victim_answer):
            alt = list(alt)
            alt_spans = _spans(alt)                        # spans of the unfiltered answer
            alt_words = drop_answer_stop_words(alt, stop_words)
            if len(alt_words) != len(hit_words) or same:
                continue
            if any(alt_spans[i][1] - alt_spans[i][0] != hit_len[i] for i in range(len(alt_words))):
                continue
            positions = sorted(hit_pos[:len(alt_words)], reverse=True)
            swapped = list(para)
            for wpos, w in zip(positions, alt_words):
                swapped[wpos] = w
            alt_ids, _ = encode(swapped, flavor, tail, max_len)
            label_sets.append(_label_row(n_labels, spans, positions, alt_ids))
    labels = label_sets[0] if len(label_sets) == 1 else label_sets
    return MlmTask(old_alg=0, text_ids_mlm=ids_mlm, text_mask_mlm=mask_mlm, mlm_labels=labels, words_mlm=masked,
                   mask_pos=hit_pos[-1], n_masked_pieces=hit_len[-1], flavor=flavor, tail=tuple(tail), max_len=max_len)


def apply_substitutions(words_mlm: List[Word], ops: Sequence[Tuple[int, int]]) -> List[Word]:
    """``update_mlm_text`` (adv_attack.py:334-341): every one-piece word of the paraphrase that equals a word the
    question just lost is replaced by the question's new word.  ``ops``: (old id, new id) pairs.  In place."""
    for old, new in ops:
        for i, w in enumerate(words_mlm):
            if w == (old,):
                words_mlm[i] = (new,)
    return words_mlm


def live_label_rows(labels):
    """Positions of a batch's MLM labels that are a target in at least one label set, as a rectangular index.

    ``labels``: int64 tensor (B, L) or (B, K, L) with ``IGNORE`` everywhere except the [MASK]-ed answer pieces
    (adv_attack.py:433-558) -- typically 1-3 of the L positions.  Returns ``(rows (B, W) int64, compact labels (B, W) or
    (B, K, W))`` with W = the largest number of live positions of a sample: a sample's live positions in ascending
    order, padded with dead positions of the same sample (whose labels are all ``IGNORE``, so they add nothing to the
    loss and the cross-entropy kernel does not even load their logits).  A white box that evaluates its MLM head on
    ``states[b, rows[b]]`` only, against the compact labels, computes the same per-sample losses and gradients as the
    dense (B, L, V) form without producing the B x L x V logits and their gradient.  One host read (W)."""
    import torch
    live = (labels != IGNORE) if labels.dim() == 2 else (labels != IGNORE).any(dim=1)
    width = max(int(live.sum(dim=1).max().item()), 1) if live.numel() else 1
    order = torch.argsort((~live).to(torch.int32), dim=1, stable=True)[:, :width]          # live first, ascending
    if labels.dim() == 2:
        return order, torch.gather(labels, 1, order)
    return order, torch.gather(labels, 2, order.unsqueeze(1).expand(-1, labels.shape[1], -1))
