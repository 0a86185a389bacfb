"""Attack-success bookkeeping and the only collective of the path (rows a12 / e of SURVEY.md section 8).

Reference: ``acc_list.append(1 if answer changed else 0)`` per sample and a running
``print('attack_accuracy', sum/len)`` (``ALBEF_attack/adv_attack.py:717-733``, ``vlmo_module.py:2063-2091``); the
reference runs single-process.  Here samples are sharded over ranks (``shard_indices``: interleaved, because the cost of
a sample grows with its number of substitutable words) and the per-rank success bits are exchanged with ONE
``all_gather`` (RCCL over xGMI on the GPU node, gloo in the CPU tests).  Payload is <= ceil(N / world) bytes per rank,
so the collective is latency-bound; no image-sized tensor ever crosses a link.  A sharded sweep gathers its second
output, the adversarial token ids of every sample, with one more small all-gather (``all_gather_rows``).
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

    def _counts(self, n_total):
        """Per-rank sample counts of the interleaved sharding -- every rank can compute them, nothing is exchanged."""
        return [len(range(r, n_total, self.world)) for r in range(self.world)]

    def all_gather_bits(self, n_total=None):
        """(bits, ids) of ALL ranks, rank-major; without recorded sample ids a rank's bits are taken to be those of its
        interleaved shard in order (single process without a collective: ids is None then).

        ONE collective: every rank contributes ``ceil(n_total / world)`` int64 words ``2 * sample id + bit`` (-1 = padding;
        ranks hold different counts when ``world`` does not divide ``n_total``).  ``n_total`` = number of samples of the
        whole sweep, sharded ``rank::world`` (``shard_indices``): the per-rank counts follow from it, so no count
        exchange is needed; a rank whose contribution does not have its expected count raises on every rank."""
        bits = self.local_bits()
        ids = torch.cat(self._ids) if self._ids and len(self._ids) == len(self._bits) else None
        if self.world == 1 and not self.force_collective:
            return bits, ids
        if n_total is None:
            if self.world > 1:
                raise ValueError("all_gather_bits(n_total=...) is required when world > 1")
            n_total = bits.numel()
        counts = self._counts(n_total)
        if bits.numel() != counts[self.rank]:
            raise RuntimeError("rank {} recorded {} success bits, its shard of {} samples has {}".format(
                self.rank, bits.numel(), n_total, counts[self.rank]))
        cap = max(max(counts), 1)
        own = ids if ids is not None else torch.arange(self.rank, n_total, self.world, device=self.device)
        word = torch.full((cap,), -1, dtype=torch.int64, device=self.device)
        word[:bits.numel()] = own * 2 + bits.to(torch.int64)
        parts = [torch.empty_like(word) for _ in range(self.world)]
        dist.all_gather(parts, word)
        self.collectives += 1
        got = torch.cat([p[:c] for p, c in zip(parts, counts)])
        if bool((got < 0).any()) or any(bool((p[c:] >= 0).any()) for p, c in zip(parts, counts)):
            raise RuntimeError("a rank contributed a different number of success bits than its shard holds")
        return (got & 1).to(torch.uint8), got >> 1

    def all_gather_rows(self, sample_ids, rows, n_total):
        """Gather one int64 row per sample from all ranks (the adversarial token ids the reference dumps at the end of
        the sweep, adv_attack.py:734-735 / vlmo_module.py:2095-2097): ONE padded all-gather of ``(cap, 1 + L)`` words
        ``[sample id, row...]``.  Returns ``(ids (n_total,), rows (n_total, L))`` sorted by sample id, on every rank."""
        ids = torch.as_tensor(sample_ids, dtype=torch.int64, device=self.device).reshape(-1)
        rows = rows.to(self.device, torch.int64).reshape(ids.numel(), rows.shape[-1])   # (0, L) stays (0, L): an empty shard
        if self.world == 1 and not self.force_collective:
            order = torch.argsort(ids)
            return ids[order], rows[order]
        counts = self._counts(n_total)
        if ids.numel() != counts[self.rank]:
            raise RuntimeError("rank {} holds {} rows, its shard of {} samples has {}".format(
                self.rank, ids.numel(), n_total, counts[self.rank]))
        cap = max(max(counts), 1)
        buf = torch.full((cap, 1 + rows.shape[1]), -1, dtype=torch.int64, device=self.device)
        buf[:ids.numel(), 0] = ids
        buf[:ids.numel(), 1:] = rows
        parts = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(parts, buf)
        self.collectives += 1
        got = torch.cat([p[:c] for p, c in zip(parts, counts)])
        order = torch.argsort(got[:, 0])
        got = got[order]
        if got.shape[0] != n_total or not torch.equal(got[:, 0], torch.arange(n_total, device=self.device)):
            raise RuntimeError("the gathered rows do not cover every sample id exactly once")
        return got[:, 0], got[:, 1:]

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

    def all_gather_rate(self, n_total=None):
        """Attack success rate over every rank's samples (the reference's final ``sum(acc_list)/len(acc_list)``)."""
        bits, _ = self.all_gather_bits(n_total)
        return float(bits.float().mean().item()) if bits.numel() else None
