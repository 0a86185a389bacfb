"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the black-box scorers that decide every attack-success bit (row f2).

  * ``rank_answer`` ........ ``ALBEF_attack/models/model_vqa.py:149-203``: first-token top-k over the answer list, then
    full-sequence re-ranking of the k candidates with the answer decoder's LM loss (``BertLMHeadModel`` with
    ``reduction='none'``, ``models/xbert.py:1265-1271``: shifted per-token cross entropy summed over the sequence),
    restated question by question with the reference's ``index_select`` / ``tile`` structure;
  * ``albef_predict`` ...... what ``Adv_attack.evaluate`` reads from it (``adv_attack.py:720-726``): per question
    ``_, pred = topk_prob.max(dim=0); answer_list[topk_id[pred]]``;
  * ``vlmo_predict`` ....... ``vqa_test_step_after_pgd`` (``vlmo/modules/objectives.py:812-829``): argmax of the answer
    classifier's logits.

**Pinned**: ``rank_answer`` by the ``rank_*`` arrays of ``tests/golden/text_golden.npz`` (the reference's own method,
compiled from its source and executed over this repository's tiny ALBEF decoder).  ``vlmo_predict`` is an argmax.
The frozen networks themselves are this repository's (model arithmetic is out of scope).
"""
import torch
import torch.nn.functional as F


def lm_loss_none(logits, labels):
    """BertLMHeadModel's loss with reduction='none' (xbert.py:1265-1271): shift, per-token CE, sum per sequence."""
    shifted = logits[:, :-1, :].contiguous()
    lab = labels[:, 1:].contiguous()
    loss = F.cross_entropy(shifted.view(-1, shifted.shape[-1]), lab.view(-1), reduction="none")
    return loss.view(logits.size(0), -1).sum(1)


def tile(x, dim, n_tile):
    """model_vqa.py:205-211."""
    init_dim = x.size(dim)
    repeat_idx = [1] * x.dim()
    repeat_idx[dim] = n_tile
    x = x.repeat(*repeat_idx)
    order = torch.cat([init_dim * torch.arange(n_tile) + i for i in range(init_dim)]).long()
    return torch.index_select(x, dim, order.to(x.device))


def rank_answer(decode, question_states, question_atts, answer_ids, answer_atts, k, pad_id=0):
    """``decode(ids, atts, states, state_atts) -> LM logits (n, L, V)`` is the answer decoder's trunk + head."""
    num_ques = question_states.size(0)
    start_ids = answer_ids[0, 0].repeat(num_ques, 1)
    logits = decode(start_ids, torch.ones_like(start_ids), question_states, question_atts)[:, 0, :]
    answer_first_token = answer_ids[:, 1]
    prob_first_token = F.softmax(logits, dim=1).index_select(dim=1, index=answer_first_token)
    topk_probs, topk_ids = prob_first_token.topk(k, dim=1)
    input_ids, input_atts = [], []
    for _b, topk_id in enumerate(topk_ids):
        input_ids.append(answer_ids.index_select(dim=0, index=topk_id))
        input_atts.append(answer_atts.index_select(dim=0, index=topk_id))
    input_ids = torch.cat(input_ids, dim=0)
    input_atts = torch.cat(input_atts, dim=0)
    targets_ids = input_ids.masked_fill(input_ids == pad_id, -100)
    q_states = tile(question_states, 0, k)
    q_atts = tile(question_atts, 0, k)
    answer_loss = lm_loss_none(decode(input_ids, input_atts, q_states, q_atts), targets_ids)
    answer_loss = answer_loss.view(input_ids.size(0), -1)
    topk_probs = topk_probs.view(-1, 1)
    log_probs = torch.cat([topk_probs.log(), -answer_loss], dim=1)
    log_probs_sum = log_probs.sum(1).view(num_ques, k)
    topk_probs = F.softmax(log_probs_sum, dim=-1)
    topk_probs, rerank_id = topk_probs.topk(k, dim=1)
    topk_ids = torch.gather(topk_ids, 1, rerank_id)
    return topk_ids, topk_probs


def albef_predict(topk_ids, topk_probs):
    """adv_attack.py:722-726, one question at a time: index (into the answer list) of the most probable answer."""
    out = []
    for topk_id, topk_prob in zip(topk_ids, topk_probs):
        _, pred = topk_prob.max(dim=0)
        out.append(int(topk_id[pred]))
    return out


def vlmo_predict(vqa_logits):
    """objectives.py:822-823: ``vqa_logits.argmax(dim=-1)`` (answer ids; the reference then maps them to strings)."""
    return [int(p) for p in vqa_logits.argmax(dim=-1)]
