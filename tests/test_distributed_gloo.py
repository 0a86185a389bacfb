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
        bits, ids = ledger.all_gather_bits()
        rate = ledger.all_gather_rate()
        torch.save({"bits": bits, "ids": ids, "rate": rate}, os.path.join(out_dir, "r{}.pt".format(rank)))
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
        assert abs(got["rate"] - want) < 1e-6


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
