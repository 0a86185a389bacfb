"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the text side of the joint attack (rows a10/a11 of SURVEY.md 8a).

Restated from source text (the orchestrators cannot be imported: they need tensorflow_hub, timm, sacred ...), so
this part of the oracle is pinned by no reference-generated vector: **parity unpinned** for a10/a11 beyond the
hand-computed schedule values in tests/test_schedule.py (which follow adv_attack.py:229-239 literally).

  * iter_schedule ......... ALBEF_attack/adv_attack.py:229-239 (= vlmo_module.py:1545-1556)
  * bert_embeddings ....... ALBEF_attack/models/xbert.py:189-216 (HF BertEmbeddings, eval mode -> no dropout)
  * dir_sim ............... ALBEF_attack/adv_attack.py:325-333 (= vlmo_module.py:1632-1640)
  * candidate_scores ...... ALBEF_attack/adv_attack.py:272-298 for position-preserving single-token substitutions
"""
import torch
import torch.nn.functional as F


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
