"""A/B of the fused L-inf step kernel: a previous build of the library against the tuning build's knob settings.

    python tools/step_ab.py <previous libvqattack_hip.so> <libvqattack_hip_tuning.so> [--batch 64 256] [--rounds 5]

Warm (20 back-to-back launches, ping-pong buffers) and cold (each launch after a 1 GiB fill: what the PGD loop sees),
medians over interleaved rounds in one process; 16 B per element.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip  # noqa: E402
from microbench_step import cold, setup, warm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("previous")
    ap.add_argument("tuning")
    ap.add_argument("--batch", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    prev, tune = _hip.load_library(os.path.abspath(args.previous)), _hip.load_library(os.path.abspath(args.tuning))
    tune.vqa_set_option.restype, tune.vqa_set_option.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int]
    flush = torch.empty(256 * 1024 * 1024, device="cuda")
    variants = [("previous build", None)] + [("this build: {} workgroups per CU, unroll {}".format(b, u), (b, u))
                                             for b, u in ((8, 4), (6, 4), (5, 4), (12, 4), (8, 2), (10, 2), (12, 2), (16, 2))]
    for batch in args.batch:
        x0, bufs, g = setup(batch)
        nbytes = 16 * x0.numel()
        res = {name: {"warm": [], "cold": []} for name, _ in variants}
        for _ in range(args.rounds):
            for name, knobs in variants:
                if knobs is None:
                    _hip._lib = prev
                else:
                    _hip._lib = tune
                    assert tune.vqa_set_option(0, knobs[0]) == 0 and tune.vqa_set_option(2, knobs[1]) == 0
                res[name]["warm"].append(warm(x0, bufs, g))
                res[name]["cold"].append(cold(x0, bufs, g, flush))
        for name, _ in variants:
            w, c = statistics.median(res[name]["warm"]), statistics.median(res[name]["cold"])
            print(json.dumps(dict(batch=batch, variant=name, warm_us=round(w, 2), warm_of_8TBs=round(nbytes / w / 8e6, 3),
                                  cold_us=round(c, 2), cold_of_8TBs=round(nbytes / c / 8e6, 3))), flush=True)
        del x0, bufs, g


if __name__ == "__main__":
    main()
