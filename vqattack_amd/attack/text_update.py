"""Word substitution between PGD blocks (row a10 of SURVEY.md section 8a), batched over samples.

Reference: ``cal_text_attack_list`` (candidate words: top-5 MLM predictions per attackable word, score threshold 0.3,
original word / sub-words / stop words filtered, ``ALBEF_attack/adv_attack.py:240-264`` + ``get_substitues`` :191-207),
``update_adv_text`` (:265-324: score every candidate by the cosine between its embedding direction and the text-embedding
gradient, sort descending, greedily accept at most one candidate per word while the sentence similarity to the ORIGINAL
question exceeds a threshold that starts at 0.95 and rises to each accepted similarity) and ``dir_sim`` (:325-333);
the VLMO copies are ``vlmo_module.py:1531-1581, 1632-1702``.

Working on token ids instead of strings: the reference only substitutes words that are a single word-piece with a
candidate that is a single word-piece (:226, :252), so a substitution never moves any other token and the sentence
re-embedding of :284-295 reduces to the position-wise embedding the scoring kernel computes (``ops.cand_dir_sim``).
The sentence-similarity model (TF-Hub Universal Sentence Encoder in the reference, :315-318) is an injectable callback:
``similarity_fn(ori_ids: list[int], new_ids: list[int]) -> float``.
"""
import numpy as np
import torch

from .. import ops

TOPK = 5                 # adv_attack.py:244
SCORE_THRESHOLD = 0.3    # get_substitues(threshold=0.3), adv_attack.py:191
SIM_THRESHOLD = 0.95     # adv_attack.py:303


def propose_candidates(mlm_logits, text_ids, attackable, banned=None, topk=TOPK, threshold=SCORE_THRESHOLD):
    """Per sample: ``[(position, [candidate ids...]), ...]`` for every attackable position with >= 1 candidate.

    ``mlm_logits`` (B, L, V) device, ``text_ids`` (B, L), ``attackable`` bool (B, L), ``banned`` optional bool (V,)
    marking ids that may never be proposed (sub-word pieces, stop words).  One device top-k, one host transfer.
    """
    logits = mlm_logits.detach()
    if banned is not None:
        logits = logits.masked_fill(banned.to(logits.device)[None, None, :], float("-inf"))
    scores, ids = torch.topk(logits, topk, dim=-1)
    scores, ids = scores.cpu().numpy(), ids.cpu().numpy()
    tid, att = text_ids.cpu().numpy(), attackable.cpu().numpy().astype(bool)
    proposals = []
    for s in range(tid.shape[0]):
        per = []
        for p in np.nonzero(att[s])[0]:
            keep = []
            for sc, v in zip(scores[s, p], ids[s, p]):
                if threshold != 0 and sc < threshold:
                    break                                  # sorted: everything after is below the threshold too
                if int(v) == int(tid[s, p]):
                    continue                               # "filter out original word"
                keep.append(int(v))
            if keep:
                per.append((int(p), keep))
        proposals.append(per)
    return proposals


def score_candidates(tables, e_ori, text_grad, proposals):
    """``dir_sim`` of every proposed (sample, position, id) in ONE kernel launch.

    ``text_grad`` is the full (B, L, D) text-embedding gradient (the probe step is run with
    ``attack_mask = range(L)``), so the gradient row of a candidate is its token position.
    Returns ``(cand int array (n, 4), scores float array (n,))`` on the host.
    """
    rows = [(s, p, p, v) for s, per in enumerate(proposals) for (p, vs) in per for v in vs]
    if not rows:
        return np.zeros((0, 4), dtype=np.int32), np.zeros((0,), dtype=np.float32)
    cand = torch.tensor(rows, dtype=torch.int32, device=e_ori.device)
    scores = ops.cand_dir_sim(tables["word"], tables["pos"], tables["type_emb"], tables["gamma"], tables["beta"],
                              tables["ln_eps"], e_ori.contiguous(), text_grad.contiguous(), cand)
    return np.asarray(rows, dtype=np.int32), scores.cpu().numpy()


def greedy_accept(cand, scores, ori_ids, cur_ids, similarity_fn, sim_threshold=SIM_THRESHOLD):
    """Sequential acceptance of ``update_adv_text`` (:300-323) for every sample; returns (new ids array, op lists).

    ``ori_ids`` / ``cur_ids``: int arrays (B, L) (original question, current adversarial question).
    """
    new_ids = np.array(cur_ids, copy=True)
    ops_per_sample = [[] for _ in range(new_ids.shape[0])]
    for s in range(new_ids.shape[0]):
        mine = np.nonzero(cand[:, 0] == s)[0] if len(cand) else []
        # python's sort is stable and `reverse=True` keeps ties in original order, like sorted(..., reverse=True)
        order = sorted(mine, key=lambda k: scores[k], reverse=True)
        taken, thr = set(), sim_threshold
        ori = [int(t) for t in ori_ids[s]]
        for k in order:
            p, v = int(cand[k, 1]), int(cand[k, 3])
            if p in taken:
                continue
            trial = new_ids[s].copy()
            trial[p] = v
            sim = float(similarity_fn(ori, [int(t) for t in trial]))
            if sim > thr:
                thr = sim
                taken.add(p)
                ops_per_sample[s].append((p, int(new_ids[s, p]), v))
                new_ids[s] = trial
    return new_ids, ops_per_sample


class BagOfEmbeddingsSimilarity:
    """Synthetic stand-in for the sentence encoder: cosine of mean token embeddings from a fixed random table.
    (There is no TensorFlow / TF-Hub on the GPU box; the real encoder plugs in through the same callable.)"""

    def __init__(self, vocab=30522, dim=64, seed=0):
        self.table = np.random.RandomState(seed).standard_normal((vocab, dim)).astype(np.float32)

    def __call__(self, ori_ids, new_ids):
        a = self.table[[t for t in ori_ids if t != 0]].mean(0)
        b = self.table[[t for t in new_ids if t != 0]].mean(0)
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-12))
