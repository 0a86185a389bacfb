"""Sharded sweep over a (synthetic) question set: bucket by schedule -> batch -> attack -> score -> gather ASR.

Mirrors what the reference's outer loops do one sample at a time (``Adv_attack.evaluate`` ``adv_attack.py:415-735``;
``VLMo.test_step`` / ``test_epoch_end`` ``vlmo_module.py:1725-2124``): skip nothing, attack, keep the adversarial text,
optionally save ``<qid>.pt`` (``adv_attack.py:714``), score with the black box, report the running attack accuracy.
Offline there are no VQAv2 images, tokenizer or checkpoints, so samples are synthetic:
images U(-1,1), questions ``[CLS] body [SEP] pad`` with 4..12 body tokens, all body tokens single-piece words.
"""
import time

import numpy as np
import torch

from .asr import SuccessLedger, shard_indices
from .runner import AttackConfig, BatchedVQAttack
from .schedule import bucket_by_schedule, gradient_steps


def synthetic_mlm_tasks(ids, dual_every, flavor, seed=0, max_len=None):
    """Synthetic stand-in for the per-question side data of the reference (victim answer, correct-answer list,
    declarative paraphrase: ``vilt_ans_table``, ``all_correct_ans``, ``chatgpt`` tables, adv_attack.py:62-79), fed through
    the real decision logic ``mlm_task.build_mlm_task`` (adv_attack.py:433-558).

    Paraphrase of sample s = its question words followed by one answer word.  One sample in ``dual_every`` -- exactly
    ceil(n / dual_every) of them, spread over the set by a seeded permutation, NOT every dual_every-th index: with the
    sweep's interleaved ``rank::world`` shards a periodic pattern would put every dual-loss sample of a 5k sweep on two
    of eight ranks -- has a victim answer that IS that word (-> old_alg = 0: the word is [MASK]-ed and becomes the MLM
    label; every third of those has a second correct answer of the same piece count -> 3-d labels); the other samples'
    victim answer does not occur in the paraphrase (-> old_alg = 1, feature loss only).  Returns one ``MlmTask`` per
    sample."""
    from . import mlm_task
    r = np.random.RandomState(seed + 1)
    slot = np.random.RandomState(seed + 7).permutation(ids.shape[0])      # sample -> its slot in the dual pattern
    tasks = []
    for s in range(ids.shape[0]):
        body = [(int(t),) for t in ids[s].tolist() if t not in (0, 101, 102)]
        in_para, other, alt = (int(v) for v in r.randint(1000, 30522, 3))
        para = body + [(in_para,)]
        is_dual = bool(dual_every) and int(slot[s]) % dual_every == 0
        answer = [(in_para,)] if is_dual else [(other,)]
        correct = [answer, [(alt,)]] if (is_dual and (int(slot[s]) // dual_every) % 3 == 0) else [answer]
        tasks.append(mlm_task.build_mlm_task(answer, correct, [True] + [False] * (len(correct) - 1), para, [], flavor,
                                             max_len=max_len))
    return tasks


def synthetic_questions(n_samples, text_len, seed=0, min_words=4, max_words=12, joint=True):
    """ids (n, L) int64, masks, attackable (n, L) bool -- on the host (tiny)."""
    r = np.random.RandomState(seed)
    ids = np.zeros((n_samples, text_len), dtype=np.int64)
    att = np.zeros((n_samples, text_len), dtype=bool)
    hi = min(max_words, text_len - 2)
    for s in range(n_samples):
        n = int(r.randint(min(min_words, hi), hi + 1))
        ids[s, 0] = 101
        ids[s, 1:1 + n] = r.randint(1000, 30522, n)
        ids[s, 1 + n] = 102
        if joint:
            att[s, 1:1 + n] = True
    return torch.from_numpy(ids), torch.from_numpy((ids != 0).astype(np.int64)), torch.from_numpy(att)


def synthetic_images(qids, image_size, device):
    out = torch.empty(len(qids), 3, image_size, image_size, device=device)
    for i, q in enumerate(qids):
        g = torch.Generator(device=device).manual_seed(1_000_003 * int(q) + 17)
        out[i].uniform_(-1, 1, generator=g)
    return out


def run_sweep(flavor, white, black, adapters, n_samples, batch, image_size, text_len, device, rank=0, world=1,
              config=None, joint=True, save_dir=None, log_every=50, seed=0, max_words=12, dual_every=0, mixed=False,
              attack=None, force_collective=False, collective_device=None, progress=None, source=None,
              mlm_logits_fn=None, banned_ids=None):
    """Returns ``dict(asr, n_total, n_local, seconds, examples_per_sec_local, gradient_steps, n_batches, mean_batch,
    global_steps, batch_global_steps, gather_seconds, adv_text, collectives, input_seconds, input_blocked_seconds,
    writer_seconds)`` on every rank: ``seconds`` is this rank's attack + scoring time (device drained, the ``.pt`` writer
    joined), ``gather_seconds`` the time of the two small all-gathers that follow (success bits, adversarial text)
    including the wait for the slowest rank -- on a sharded sweep that wait IS the shard imbalance; ``global_steps`` the
    white-box forward + backward passes this rank ran (a mixed batch runs as many as its longest sample needs),
    ``gradient_steps`` the sum over its samples of each sample's own gradient steps; ``input_seconds`` the host time spent
    producing image batches (``source.images``), ``input_blocked_seconds`` the part of it spent waiting for decoded pixels,
    ``writer_seconds`` the wait for the asynchronous ``.pt`` writer after the last batch.
    ``progress(done, n_local)``: called after every batch (bench.py: a line per minute for the GPU box's liveness check).
    ``attack``: a ready ``BatchedVQAttack`` (or an object with its ``attack_batch`` / ``attack_mixed`` / ``cfg``) instead
    of one built from ``adapters`` -- the multi-rank CPU tests inject a stand-in to exercise shard -> ledger -> gather.
    ``source``: where the pairs come from (``attack/dataset.py``: ``VqaFilePairs`` = the reference's annotation json +
    image files + in-tree tables; default ``SyntheticPairs(n_samples, ...)``); samples are addressed by their index in the
    source, files and the adversarial-text json are named by the source's question ids.

    ``mixed=False``: samples are bucketed by (schedule, loss mode) and every batch is schedule-pure
    (``BatchedVQAttack.attack_batch``).  ``mixed=True``: ONE bucket -- samples are batched in index order whatever their
    word counts and loss modes (``attack_mixed``: prefix scheduling, dual-loss samples alternate feature and MLM steps
    inside the shared white-box pass)."""
    if source is None:
        from .dataset import SyntheticPairs
        source = SyntheticPairs(n_samples, text_len, image_size, flavor, seed=seed, joint=joint, max_words=max_words,
                                dual_every=dual_every)
    n_samples = source.n
    ids, masks, att, tasks = source.ids, source.masks, source.attackable, source.tasks
    text_len = ids.shape[1]
    dual = torch.tensor([t.old_alg == 0 for t in tasks], dtype=torch.bool)
    mine = shard_indices(n_samples, rank, world)
    device = torch.device(device)
    if attack is None:
        attack = BatchedVQAttack(adapters, flavor, white.embedding_tables(), config or AttackConfig(),
                                 banned_ids=banned_ids)
    # collective_device: where the gathered tensors live -- the compute device under RCCL (default); the host when several
    # ranks rehearse on ONE GPU over gloo (bench.py, VQA_DIST_BACKEND=gloo)
    ledger = SuccessLedger(world, rank, collective_device if collective_device is not None else device,
                           force_collective=force_collective)   # True: a 1-rank torchrun launch
    # one bucket per (schedule, loss mode): a batch shares its block structure and its old_alg;
    # with mixed=True all samples share ONE bucket (key -2) and are scheduled per sample inside the batch
    buckets = bucket_by_schedule([(-2 if mixed else int(att[i].sum()) * 2 + int(dual[i])) for i in mine])
    writer = None
    if save_dir:
        from ..preprocess import AdvImageWriter
        writer = AdvImageWriter(save_dir, device)     # <qid>.pt, (1,3,H,W) fp32, like adv_attack.py:714
    adv_rows, adv_index = [], []
    steps = 0
    done = 0
    n_batches = 0
    batch_global_steps = []
    plan = []                                          # [(bucket key, sample indices of the batch)] in attack order
    for key, local in buckets.items():
        if key == -2:
            # a mixed batch runs as many global steps as its longest sample needs and finished samples leave the white-box
            # batch: batches of similar length waste the fewest slots.  Samples are independent, so the order is free
            # (the ledger records sample ids); equal lengths keep the loss modes together.
            local = sorted(local, key=lambda j: (-int(att[mine[j]].sum()), bool(dual[mine[j]]), j))
        for lo in range(0, len(local), batch):
            plan.append((key, [mine[j] for j in local[lo:lo + batch]]))
    t0 = time.perf_counter()
    if plan:
        source.prefetch(plan[0][1])
    for at, (key, index) in enumerate(plan):
        n_words, is_dual = key // 2, bool(key % 2)
        images = source.images(index, device)
        if at + 1 < len(plan):
            source.prefetch(plan[at + 1][1])          # the next batch's files are read while this one is attacked
        tid, tmask, tatt = ids[index].to(device), masks[index].to(device), att[index].to(device)
        clean = black.vqa_answer(images, tid, tmask)
        if key == -2:
            batch_tasks = [tasks[q] for q in index]
            res = attack.attack_mixed(images, tid, tmask, tatt, mlm_logits_fn=mlm_logits_fn,
                                      tasks=batch_tasks if any(t.old_alg == 0 for t in batch_tasks) else None)
        elif is_dual:
            res = attack.attack_batch(images, tid, tmask, tatt, mlm_logits_fn=mlm_logits_fn, dual=True,
                                      tasks=[tasks[q] for q in index])
        else:
            res = attack.attack_batch(images, tid, tmask, tatt, mlm_logits_fn=mlm_logits_fn)
        after = black.vqa_answer(res.adv_images, res.adv_text_ids, tmask)
        ledger.record(after != clean, sample_ids=index)
        # per-sample gradient steps, the same quantity on the bucketed and on the mixed path
        steps += res.sample_steps or res.gradient_steps * (1 if key == -2 else len(index))
        batch_global_steps.append(int(res.global_steps or (res.gradient_steps if key != -2 else 0)))
        if key != -2:
            assert res.gradient_steps == gradient_steps(n_words, attack.cfg.budget)
        adv_rows.append(res.adv_text_ids[:, :text_len])        # stays on the device until the sweep's one gather
        adv_index += index
        if writer is not None:
            writer.write(res.adv_images, [source.qids[i] for i in index])
        done += len(index)
        n_batches += 1
        if progress is not None:
            progress(done, len(mine))
        if rank == 0 and log_every and done % log_every < len(index):
            bits = ledger.local_bits()
            print("attack_accuracy", float(bits.float().mean().item()), "({} local samples)".format(done),
                  flush=True)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    writer_dt = 0.0
    if writer is not None:
        tw = time.perf_counter()
        writer.close()                                # every <qid>.pt of this rank is on disk when the clock stops
        writer_dt = time.perf_counter() - tw
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    asr = ledger.all_gather_rate(n_samples)
    # the sweep's second output: every sample's adversarial question on every rank (rank 0 writes the json the reference
    # writes at the end, adv_attack.py:734-735 / vlmo_module.py:2095-2097) -- one small all-gather of (index, ids) rows
    rows = torch.cat(adv_rows) if adv_rows else torch.zeros(0, text_len, dtype=torch.int64, device=device)
    all_q, all_rows = ledger.all_gather_rows(adv_index, rows, n_samples)
    all_q, all_rows = all_q.cpu(), all_rows.cpu()          # drains the gathers
    gather_dt = time.perf_counter() - t1
    adv_text = {str(source.qids[q]): row for q, row in zip(all_q.tolist(), all_rows.tolist())}
    input_dt, blocked_dt = float(getattr(source, "seconds_images", 0.0)), float(getattr(source, "seconds_blocked", 0.0))
    source.close()
    return dict(asr=asr, n_total=n_samples, n_local=len(mine), seconds=dt,
                examples_per_sec_local=len(mine) / dt if dt > 0 else None, gradient_steps=steps, adv_text=adv_text,
                n_batches=n_batches, mean_batch=(len(mine) / n_batches if n_batches else 0.0),
                global_steps=sum(batch_global_steps), batch_global_steps=batch_global_steps, gather_seconds=gather_dt,
                n_dual_local=int(sum(bool(dual[i]) for i in mine)), input_seconds=input_dt,
                input_blocked_seconds=blocked_dt, writer_seconds=writer_dt, skipped=int(getattr(source, "skipped", 0)),
                collectives=ledger.collectives)
