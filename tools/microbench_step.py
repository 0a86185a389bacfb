"""Interleaved A/B sweep of the fused L-inf step kernel's tuning knobs on one MI355X.

    python tools/microbench_step.py [--batch 64 256] [--rounds 5]

Two regimes per variant, medians over interleaved rounds in ONE process (cdna_hip_programming.md, rule 24):
  warm: 20 back-to-back launches, ping-ponging x/out (what a micro-benchmark sees);
  cold: every launch preceded by a 1 GiB fill that evicts L2 / Infinity Cache (what the PGD loop sees: a full
        white-box forward + backward runs between two launches), timed per launch with HIP events.
"""
import argparse
import itertools
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip, ops  # noqa: E402


def setup(batch):
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    g.view(-1)[::1000] = 0
    return x0, [x, torch.empty_like(x)], g


def set_knobs(bpc, nt, unroll, chunked):
    for opt, val in ((0, bpc), (1, nt), (2, unroll), (3, chunked)):     # needs the tuning build (VQA_TUNING_LIB=1)
        if not _hip.set_option(opt, val):
            raise SystemExit("the knob sweep needs `python -m vqattack_amd.build --tuning` and VQA_TUNING_LIB=1")


def warm(x0, bufs, g, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.linf_step(bufs[0], g, x0, 0.01, 0.125, -1, 1, out=bufs[1])
    a.record()
    for i in range(reps):
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def cold(x0, bufs, g, flush, reps=4):
    ts = []
    for i in range(reps):
        flush.fill_(float(i))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    flush = torch.empty(256 * 1024 * 1024, device="cuda")   # 1 GiB
    variants = [(bpc, nt, u, ch) for bpc, nt, u, ch in itertools.product((4, 8, 16), (1, 3), (2, 4, 8), (0, 1))]
    for batch in args.batch:
        x0, bufs, g = setup(batch)
        nbytes = 16 * x0.numel()
        res = {v: {"warm": [], "cold": []} for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                set_knobs(*v)
                res[v]["warm"].append(warm(x0, bufs, g))
                res[v]["cold"].append(cold(x0, bufs, g, flush))
        rows = []
        for v in variants:
            w, c = statistics.median(res[v]["warm"]), statistics.median(res[v]["cold"])
            rows.append(dict(batch=batch, blocks_per_cu=v[0], nt=v[1], unroll=v[2], chunked=v[3], warm_us=round(w, 2),
                             warm_GBs=round(nbytes / w / 1e3), cold_us=round(c, 2), cold_GBs=round(nbytes / c / 1e3)))
        for r in sorted(rows, key=lambda r: r["cold_us"]):
            print(json.dumps(r), flush=True)
        del x0, bufs, g
    set_knobs(12, 13, 4, 0)        # the shipped defaults


if __name__ == "__main__":
    main()
