"""N > 1 path on CPU: world_size-2 gloo run of the sharding + success-bit all-gather (the path's only collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vqattack_amd.attack.asr import SuccessLedger, shard_indices


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_indices(n_samples, rank, world)
        ledger = SuccessLedger(world, rank, "cpu")
        # deterministic fake outcome per sample id: success iff id % 3 == 0; recorded in two uneven chunks
        cut = len(mine) // 3
        for chunk in (mine[:cut], mine[cut:]):
            if chunk:
                ledger.record(torch.tensor([i % 3 == 0 for i in chunk]), sample_ids=chunk)
        bits, ids = ledger.all_gather_bits(n_samples)
        rate = ledger.all_gather_rate(n_samples)
        assert ledger.collectives == 2, "one all-gather per call: counts are computed, not exchanged"
        torch.save({"bits": bits, "ids": ids, "rate": rate, "running": ledger.running_rate()},
                   os.path.join(out_dir, "r{}.pt".format(rank)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_samples", [11, 8, 1])
def test_two_rank_success_gather(tmp_path, n_samples):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_samples, str(tmp_path)), nprocs=world, join=True)
    want = sum(1 for i in range(n_samples) if i % 3 == 0) / n_samples
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r{}.pt".format(rank)))
        assert sorted(got["ids"].tolist()) == list(range(n_samples))        # every sample exactly once
        assert all(bool(b) == (i % 3 == 0) for b, i in zip(got["bits"].tolist(), got["ids"].tolist()))
        assert abs(got["rate"] - want) < 1e-6 and abs(got["running"] - want) < 1e-6


def test_shards_partition_the_sweep():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard_indices(5000, r, world))
        assert seen == list(range(5000))
        sizes = [len(shard_indices(5000, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_single_rank_ledger():
    led = SuccessLedger()
    assert led.all_gather_rate() is None
    led.record(torch.tensor([True, False, True, True]))
    assert led.all_gather_rate() == 0.75


# ---------------------------------------------------------------------------------------------------------------
# run_sweep's whole N > 1 flow on CPU: interleaved shards, uneven (schedule, loss-mode) buckets, ragged last batches,
# per-rank ledger, all-gather of bits + ids.  The attack and the black box are stand-ins (the real ones need the GPU);
# the success bit of a sample is a pure function of its id so that the gathered ASR has a known value.
class _FakeAttack:
    class cfg:
        budget = 40

    def __init__(self):
        self.calls = []

    def _result(self, images, text_ids, attackable, per_sample_steps):
        from vqattack_amd.attack.runner import BatchResult
        adv_ids = text_ids.clone()
        adv_ids[:, 1] = -adv_ids[:, 1]                         # mark the first body token as "substituted"
        self.calls.append(images.shape[0])
        return BatchResult(adv_images=images + 0.01, adv_text_ids=adv_ids, gradient_steps=per_sample_steps)

    def attack_batch(self, images, text_ids, text_masks, attackable, dual=False, tasks=None, **_kw):
        n_words = int(attackable[0].sum())
        assert bool((attackable.sum(dim=1) == n_words).all()), "a batch must be schedule-pure"
        if dual:
            assert tasks is not None and all(t.old_alg == 0 for t in tasks) and len(tasks) == images.shape[0]
        return self._result(images, text_ids, attackable, 40 + n_words)

    def attack_mixed(self, images, text_ids, text_masks, attackable, **_kw):
        return self._result(images, text_ids, attackable, int((40 + attackable.sum(dim=1)).sum()))


class _FakeBlack:
    """The answer changes under the fake attack iff the ORIGINAL first body token id is divisible by 3."""

    def vqa_answer(self, images, text_ids, text_masks):
        t = text_ids[:, 1]
        return ((t < 0) & ((-t) % 3 == 0)).long()


def _sweep_worker(rank, world, port, n_samples, mixed, out_dir, batch=4):
    from vqattack_amd.attack.sweep import run_sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = _FakeAttack()
        res = run_sweep("vlmo", None, _FakeBlack(), None, n_samples=n_samples, batch=batch, image_size=8, text_len=12,
                        device="cpu", rank=rank, world=world, log_every=0, max_words=5, dual_every=3, mixed=mixed,
                        attack=fake)
        torch.save({"asr": res["asr"], "n_local": res["n_local"], "steps": res["gradient_steps"],
                    "adv_text": res["adv_text"], "calls": fake.calls, "collectives": res["collectives"]},
                   os.path.join(out_dir, "s{}.pt".format(rank)))
    finally:
        dist.destroy_process_group()


def _check_gathered_text(adv_text, n_samples):
    """The complete adversarial-text output (adv_attack.py:734-735): every qid, with the ids the (fake) attack produced."""
    from vqattack_amd.attack.sweep import synthetic_questions
    ids, _, _ = synthetic_questions(n_samples, 12, seed=0, max_words=5)
    want = ids.clone()
    want[:, 1] = -want[:, 1]
    assert sorted(map(int, adv_text)) == list(range(n_samples))
    assert all(adv_text[str(q)] == want[q].tolist() for q in range(n_samples))


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("n_samples", [23, 10])
def test_two_rank_sweep_shard_ledger_gather(tmp_path, n_samples, mixed):
    from vqattack_amd.attack.sweep import synthetic_questions
    world = 2
    mp.spawn(_sweep_worker, args=(world, _free_port(), n_samples, mixed, str(tmp_path)), nprocs=world, join=True)
    ids, _, att = synthetic_questions(n_samples, 12, seed=0, max_words=5)
    want_asr = float((ids[:, 1] % 3 == 0).float().mean())
    got = [torch.load(os.path.join(str(tmp_path), "s{}.pt".format(r))) for r in range(world)]
    for r in range(world):
        _check_gathered_text(got[r]["adv_text"], n_samples)        # rank 0 (and every other rank) holds ALL samples' text
        assert abs(got[r]["asr"] - want_asr) < 1e-6                                 # same gathered rate on every rank
        assert got[r]["n_local"] == len(range(r, n_samples, world))
        assert max(got[r]["calls"]) <= 4 and sum(got[r]["calls"]) == got[r]["n_local"]   # ragged batches, none lost
        assert got[r]["collectives"] == 2                                           # success bits + adversarial text
    assert got[0]["steps"] + got[1]["steps"] == int((40 + att.sum(dim=1)).sum())


def test_sweep_with_fewer_samples_than_ranks(tmp_path):
    """A rank whose shard is EMPTY (n_samples < world) still takes part in both all-gathers with its (0, L) rows -- it must
    not raise before the collective and leave the other ranks waiting."""
    world, n_samples = 2, 1
    mp.spawn(_sweep_worker, args=(world, _free_port(), n_samples, True, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "s{}.pt".format(r))) for r in range(world)]
    assert [g["n_local"] for g in got] == [1, 0] and got[1]["calls"] == []
    for g in got:
        _check_gathered_text(g["adv_text"], n_samples)
        assert g["collectives"] == 2 and g["asr"] in (0.0, 1.0)


def test_eight_rank_sweep_of_5003_samples_gathers_bits_and_text(tmp_path):
    """BASELINE configs[3]'s sharding at its real size (5k samples over 8 ranks, uneven shards: 5003 = 8 * 625 + 3) on
    gloo with the stand-in attack: ASR over all samples and the complete adversarial-text output on rank 0."""
    from vqattack_amd.attack.sweep import synthetic_questions
    world, n_samples = 8, 5003
    mp.spawn(_sweep_worker, args=(world, _free_port(), n_samples, True, str(tmp_path), 64), nprocs=world, join=True)
    ids, _, att = synthetic_questions(n_samples, 12, seed=0, max_words=5)
    want_asr = float((ids[:, 1] % 3 == 0).float().mean())
    got = [torch.load(os.path.join(str(tmp_path), "s{}.pt".format(r))) for r in range(world)]
    _check_gathered_text(got[0]["adv_text"], n_samples)
    assert [g["n_local"] for g in got] == [626, 626, 626, 625, 625, 625, 625, 625]
    assert all(abs(g["asr"] - want_asr) < 1e-6 for g in got)
    assert sum(g["steps"] for g in got) == int((40 + att.sum(dim=1)).sum())
