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
    marking ids that may never be proposed (``##`` word pieces, stop words).  One device top-k, one host transfer.

    Order of operations as in the reference (``cal_text_attack_list`` :243-264 + ``get_substitues`` :191-207): the
    top-5 is taken over the WHOLE vocabulary, the list is cut at the first score below the threshold, and only then are
    the original word, word pieces and stop words dropped -- a banned id still occupies one of the five slots.
    """
    scores, ids = torch.topk(mlm_logits.detach(), topk, dim=-1)
    scores, ids = scores.cpu().numpy(), ids.cpu().numpy()
    tid, att = text_ids.cpu().numpy(), attackable.cpu().numpy().astype(bool)
    ban = None if banned is None else banned.cpu().numpy().astype(bool)
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
                if ban is not None and ban[int(v)]:
                    continue                               # "filter out sub-word" / filter_words
                keep.append(int(v))
            if keep:
                per.append((int(p), keep))
        proposals.append(per)
    return proposals


def substitutable_words(keys, word_is_filtered):
    """Word indices the reference may substitute (``cal_text_attack_list`` :224-228): exactly one word piece and not a
    filter word.  ``keys``: per word its ``[start, end)`` span in word-piece coordinates; token position = start + 1."""
    return [i for i, (k, f) in enumerate(zip(keys, word_is_filtered)) if k[1] - k[0] == 1 and not f]


class CandidatePlan:
    """The (sample, position, candidate id) rows of a batch's proposals: fixed for a whole attack, so they are built and
    uploaded ONCE; every substitution round only re-scores them."""

    def __init__(self, proposals, device):
        rows = [(s, p, p, v) for s, per in enumerate(proposals) for (p, vs) in per for v in vs]
        self.rows = np.asarray(rows, dtype=np.int32).reshape(-1, 4)
        self.device_rows = torch.as_tensor(self.rows, device=device) if len(rows) else \
            torch.zeros((0, 4), dtype=torch.int32, device=device)

    def __len__(self):
        return self.rows.shape[0]

    def restricted_to(self, samples):
        """Rows of the given samples only (``attack_mixed``: only the samples whose probe step fires take part)."""
        keep = np.isin(self.rows[:, 0], np.asarray(sorted(samples), dtype=np.int32)) if len(self) else \
            np.zeros(0, dtype=bool)
        sub = CandidatePlan.__new__(CandidatePlan)
        idx = np.nonzero(keep)[0]                       # host-side: the gather below has a known size (no device sync)
        sub.rows = self.rows[idx]
        sub.device_rows = self.device_rows[torch.as_tensor(idx, device=self.device_rows.device)] if len(self) else \
            self.device_rows
        return sub


def score_plan(tables, e_ori, text_grad, plan):
    """``dir_sim`` of every row of the plan in ONE launch; fp32 (n,) on the device, no host transfer.

    ``text_grad`` is the full (B, L, D) text-embedding gradient (the probe step is run with
    ``attack_mask = range(L)``), so the gradient row of a candidate is its token position."""
    if len(plan) == 0:
        return torch.zeros(0, dtype=torch.float32, device=e_ori.device)
    return ops.cand_dir_sim(tables["word"], tables["pos"], tables["type_emb"], tables["gamma"], tables["beta"],
                            tables["ln_eps"], e_ori.contiguous(), text_grad.contiguous(), plan.device_rows)


def score_candidates(tables, e_ori, text_grad, proposals):
    """Host-facing variant: ``(cand int array (n, 4), scores float array (n,))`` (one device->host transfer)."""
    plan = proposals if isinstance(proposals, CandidatePlan) else CandidatePlan(proposals, e_ori.device)
    return plan.rows, score_plan(tables, e_ori, text_grad, plan).cpu().numpy()


def greedy_accept(cand, scores, ori_ids, cur_ids, similarity_fn, sim_threshold=SIM_THRESHOLD):
    """Sequential acceptance of ``update_adv_text`` (:300-323) for every sample ON THE HOST, for an injected
    ``similarity_fn`` (a real sentence encoder); returns (new ids array, op lists in acceptance order).

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
    """Synthetic stand-in for the sentence encoder: cosine of mean token embeddings from a fixed table
    (there is no TensorFlow / TF-Hub on the GPU box; the real encoder plugs in through the same callable).

    Exposes ``device_table``: with it the whole acceptance loop runs on the device (``ops.greedy_accept`` /
    ``vqa_greedy_accept``) -- no per-candidate host call, no transfer.  ``__call__`` is the same arithmetic on the host."""

    def __init__(self, vocab=30522, dim=64, seed=0, table=None):
        self.table = np.random.RandomState(seed).standard_normal((vocab, dim)).astype(np.float32) if table is None \
            else np.ascontiguousarray(table, dtype=np.float32)
        self._dev = {}

    def device_table(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = torch.as_tensor(self.table, device=device).contiguous()
        return self._dev[key]

    def __call__(self, ori_ids, new_ids):
        a = self.table[[t for t in ori_ids if t != 0]].mean(0)
        b = self.table[[t for t in new_ids if t != 0]].mean(0)
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-12))


def accept_round(plan, scores, ori_ids, adv_ids, similarity_fn, sim_threshold=SIM_THRESHOLD):
    """One acceptance round for the batch; ``adv_ids`` (B, L) int64 device is updated IN PLACE.

    Returns ``(new_id, rank)`` int32 (B, L) device tensors: the id accepted at each position (-1 = none) and the order
    in which its sample accepted it.  With a similarity that has a ``device_table`` everything stays on the device;
    otherwise the scores and ids make one round trip and ``similarity_fn`` is called per candidate on the host."""
    dev = adv_ids.device
    if len(plan) == 0:
        none = torch.full(adv_ids.shape, -1, dtype=torch.int32, device=dev)
        return none, none.clone()
    table_fn = getattr(similarity_fn, "device_table", None)
    if table_fn is not None:
        return ops.greedy_accept(plan.device_rows, scores, ori_ids.contiguous(), adv_ids, table_fn(dev), sim_threshold)
    new_np, ops_list = greedy_accept(plan.rows, scores.cpu().numpy(), ori_ids.cpu().numpy(), adv_ids.cpu().numpy(),
                                     similarity_fn, sim_threshold)
    new_id = np.full(new_np.shape, -1, dtype=np.int32)
    rank = np.full(new_np.shape, -1, dtype=np.int32)
    for s, per in enumerate(ops_list):
        for r, (p, _old, v) in enumerate(per):
            new_id[s, p], rank[s, p] = v, r
    adv_ids.copy_(torch.as_tensor(new_np, device=dev, dtype=adv_ids.dtype))
    return torch.as_tensor(new_id, device=dev), torch.as_tensor(rank, device=dev)


def substitution_lists(prev_ids, new_id, rank):
    """Host view of one round: per sample ``[(position, old id, new id), ...]`` in acceptance order."""
    prev, new, rk = prev_ids.cpu().numpy(), new_id.cpu().numpy(), rank.cpu().numpy()
    out = []
    for s in range(new.shape[0]):
        pos = [p for p in np.nonzero(new[s] >= 0)[0]]
        pos.sort(key=lambda p: rk[s, p])
        out.append([(int(p), int(prev[s, p]), int(new[s, p])) for p in pos])
    return out
