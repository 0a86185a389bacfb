"""Attack-success bookkeeping and the only collective of the path (rows a12 / e of SURVEY.md section 8).

Reference: ``acc_list.append(1 if answer changed else 0)`` per sample and a running
``print('attack_accuracy', sum/len)`` (``ALBEF_attack/adv_attack.py:717-733``, ``vlmo_module.py:2063-2091``); the
reference runs single-process.  Here samples are sharded over ranks (``shard_indices``: interleaved, because the cost of
a sample grows with its number of substitutable words) and the per-rank success bits are exchanged with ONE
``all_gather`` (RCCL over xGMI on the GPU node, gloo in the CPU tests).  Payload is <= ceil(N / world) bytes per rank,
so the collective is latency-bound; no image-sized tensor ever crosses a link.
"""
import torch
import torch.distributed as dist


def shard_indices(n_samples, rank, world):
    """Interleaved shard ``rank::world`` (what ``DistributedSampler(shuffle=False)`` gives the reference's loaders,
    ``vlmo/datamodules/multitask_datamodule.py:54``)."""
    return list(range(rank, n_samples, world))


class SuccessLedger:
    def __init__(self, world=1, rank=0, device="cpu", force_collective=False):
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self.force_collective = force_collective     # run the all-gather even for world == 1 (1-rank torchrun)
        self.collectives = 0                         # all_gather / all_reduce calls issued so far (evidence for tests)
        self.reset()

    def reset(self):
        self._bits = []
        self._ids = []

    def record(self, success, sample_ids=None):
        """``success``: bool/uint8 tensor (n,) -- True where the black-box answer changed."""
        bits = success.to(self.device, torch.uint8).reshape(-1)
        self._bits.append(bits)
        if sample_ids is not None:
            self._ids.append(torch.as_tensor(sample_ids, device=self.device, dtype=torch.int64).reshape(-1))

    def local_bits(self):
        if not self._bits:
            return torch.zeros(0, dtype=torch.uint8, device=self.device)
        return torch.cat(self._bits)

    def all_gather_bits(self):
        """(bits, ids) of ALL ranks, rank-major; ids is None unless every record() call passed sample ids.
        Ranks may hold different counts (5000 samples over 8 ranks): counts are gathered first, buffers padded."""
        bits = self.local_bits()
        ids = torch.cat(self._ids) if self._ids and len(self._ids) == len(self._bits) else None
        if self.world == 1 and not self.force_collective:
            return bits, ids
        # one small gather carries (count, "I can supply ids") so every rank takes the same collective path
        has_ids = 1 if (ids is not None or bits.numel() == 0) else 0
        n = torch.tensor([bits.numel(), has_ids], device=self.device, dtype=torch.int64)
        meta = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(meta, n)
        self.collectives += 1
        counts = [int(m[0].item()) for m in meta]
        with_ids = all(int(m[1].item()) == 1 for m in meta)
        cap = max(counts) if counts else 0
        pad = torch.zeros(cap, dtype=torch.uint8, device=self.device)
        pad[:bits.numel()] = bits
        parts = [torch.zeros_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad)
        self.collectives += 1
        all_bits = torch.cat([p[:c] for p, c in zip(parts, counts)])
        all_ids = None
        if with_ids:
            ipad = torch.zeros(cap, dtype=torch.int64, device=self.device)
            if ids is not None:
                ipad[:ids.numel()] = ids
            iparts = [torch.zeros_like(ipad) for _ in range(self.world)]
            dist.all_gather(iparts, ipad)
            self.collectives += 1
            all_ids = torch.cat([p[:c] for p, c in zip(iparts, counts)])
        return all_bits, all_ids

    def running_rate(self):
        """Attack accuracy so far over ALL ranks: one ``all_reduce(SUM)`` of (successes, samples) -- the running
        ``print('attack_accuracy', sum(acc_list) / len(acc_list))`` of the reference (adv_attack.py:732-733) for a sharded
        sweep.  Collective: every rank must call it at the same point."""
        bits = self.local_bits()
        pair = torch.tensor([float(bits.sum().item()), float(bits.numel())], device=self.device, dtype=torch.float64)
        if self.world > 1 or self.force_collective:
            dist.all_reduce(pair, op=dist.ReduceOp.SUM)
            self.collectives += 1
        return float(pair[0] / pair[1]) if float(pair[1]) > 0 else None

    def all_gather_rate(self):
        """Attack success rate over every rank's samples (the reference's final ``sum(acc_list)/len(acc_list)``)."""
        bits, _ = self.all_gather_bits()
        return float(bits.float().mean().item()) if bits.numel() else None
