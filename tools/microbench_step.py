"""Sweep the tuning knobs of the fused L-inf step kernel on one MI355X and print achieved algorithmic GB/s.

    python tools/microbench_step.py [--batch 64 256] [--reps 50]

Timing: torch.cuda.Event on torch's current stream (the stream the kernels are launched on), `reps` back-to-back
launches per measurement, ping-ponging x/out so every launch reads and writes fresh lines.
"""
import argparse
import json

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vqattack_amd import _hip, ops


def time_step(batch, reps, blocks_per_cu, nt):
    lib = _hip.lib()
    assert lib.vqa_set_option(0, blocks_per_cu) == 0 and lib.vqa_set_option(1, nt) == 0
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    g.view(-1)[::1000] = 0
    bufs = [x, torch.empty_like(x)]
    for i in range(5):
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    nbytes = 16 * x.numel()
    return ms, nbytes / ms / 1e6   # GB/s


def time_copy(batch, reps):
    shape = (batch, 3, 384, 384)
    src = torch.randn(shape, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, 8 * src.numel() / ms / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0))
    for batch in args.batch:
        ms, gbs = time_copy(batch, args.reps)
        print(json.dumps({"kernel": "torch copy_ (8 B/elem)", "batch": batch, "ms": round(ms, 4), "GB/s": round(gbs)}))
        for bpc in (2, 4, 8, 16, 32):
            for nt in (0, 1, 2, 3):
                ms, gbs = time_step(batch, args.reps, bpc, nt)
                print(json.dumps({"kernel": "vqa_linf_step", "batch": batch, "blocks_per_cu": bpc, "nt": nt,
                                  "ms": round(ms, 4), "GB/s": round(gbs), "frac_of_8TBs": round(gbs / 8000, 3)}),
                      flush=True)


if __name__ == "__main__":
    main()
