"""Does the relative placement of the step kernel's four streams (x, grad, x0, out) matter?

All four buffers of a PGD iteration have the same size and, coming from the caching allocator, 2 MiB-aligned bases: the
four streams then walk the same low/middle address bits in lock step.  This probe carves the four buffers out of one
arena with a per-buffer skew (0, s, 2s, 3s bytes on top of the 2 MiB-aligned slots) and times back-to-back launches.

    python tools/skew_probe.py [--batch 256]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    shape = (args.batch, 3, 384, 384)
    n = args.batch * 3 * 384 * 384
    nbytes = 4 * n
    slot = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + (8 << 20)
    arena = torch.empty(4 * slot + (64 << 20), dtype=torch.uint8, device="cuda")
    base = arena.data_ptr()
    pad = (-base) % (2 << 20)
    gen = torch.Generator(device="cuda").manual_seed(0)
    src = [torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen) for _ in range(3)]
    for skew in (0, 128, 256, 1024, 4096, 4096 + 256, 65536 + 1024, (1 << 20) + 4096 + 256, (3 << 20) + 8192 + 512):
        bufs = []
        for i in range(4):
            off = pad + i * slot + i * skew
            bufs.append(arena[off:off + nbytes].view(torch.float32).view(shape))
        for b_, s_ in zip(bufs[:3], src):
            b_.copy_(s_)
        x, g, x0, out = bufs
        ms = timeit(lambda: ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out))
        print(json.dumps(dict(kernel="vqa_linf_step", batch=args.batch, skew_bytes=skew, us=round(ms * 1e3, 1),
                              GBs=round(16 * n / ms / 1e6))), flush=True)


if __name__ == "__main__":
    main()
