"""Why did bench.py's `roofline_attention` time the forward at 0.88 ms when tools/attn_bench.py (0.69 ms) and the kernel
trace inside the attack (0.68 ms) agree with each other?  One process, one box, the SAME call (`attention._forward` with
saved scores at the bench's shape), timed under the conditions that differ between the two harnesses:

  fresh        right after start-up: 2 warm-up launches + 10 timed (bench.py's round-4 harness), 3 + 10 (the tool's),
               10 + 50
  after_attack after one real attack of a batch of 64 (allocator holds tens of GB of cached blocks, card under load)
  after_idle   after 2 s of host-side sleep (what bench.py's stream probe / JSON work can amount to): clocks dropped?
  after_probe  right after the HBM-bound step microbench (bench.py's order)
  per-launch   the same launches timed one event pair per launch (shows a ramp, if there is one)

Usage: python tools/attn_harness_ab.py [--seq 591] > profiles/r05/attn_harness_ab.jsonl
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import attention  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=591)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-attack", action="store_true")
    args = ap.parse_args()
    b, h, s = args.batch, 12, args.seq
    gen = torch.Generator(device="cuda").manual_seed(2)
    qkv = torch.randn(b, s, 3, h, 64, device="cuda", generator=gen)
    store = torch.zeros(1, h, s, (s + 31) // 32 * 32, device="cuda")
    store[..., :s] = torch.randn(1, h, s, s, device="cuda", generator=gen) * 0.02
    bias = store[..., :s].expand(b, -1, -1, -1)
    bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o, lse, scores = attention._forward(q, k, v, bias, bstr, 0.125, save_scores=True)
    go = torch.randn(o.shape, device="cuda", generator=gen)
    dqkv = torch.empty_like(qkv)

    def fwd():
        attention._forward(q, k, v, bias, bstr, 0.125, save_scores=True)

    def bwd():
        attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], 0.125,
                            scores=scores)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 4), round(host / reps * 1e3, 4)

    def per_launch(fn, n):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(n)]

    def report(phase):
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for warm, reps in ((2, 10), (3, 10), (10, 50)):
                ms, host = timed(fn, warm, reps)
                print(json.dumps(dict(phase=phase, call=name, warmup=warm, reps=reps, ms=ms, host_ms_per_call=host)),
                      flush=True)
        print(json.dumps(dict(phase=phase, call="fwd", per_launch_ms=per_launch(fwd, 16))), flush=True)

    report("fresh")
    time.sleep(2.0)
    ms, host = timed(fwd, 2, 10)
    print(json.dumps(dict(phase="after_idle_2s", call="fwd", warmup=2, reps=10, ms=ms, host_ms_per_call=host)), flush=True)
    time.sleep(2.0)
    print(json.dumps(dict(phase="after_idle_2s", call="fwd", per_launch_ms=per_launch(fwd, 16))), flush=True)
    if not args.no_attack:
        from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
        white = FrozenVlmo(vlmo_base(384), seed=0).to("cuda")
        attack = BatchedVQAttack(VlmoAttackAdapters(white), "vlmo", white.embedding_tables(), AttackConfig(budget=40))
        ids = torch.zeros(b, 40, dtype=torch.long, device="cuda")
        ids[:, 0], ids[:, 1:13], ids[:, 13] = 101, torch.randint(1000, 30522, (b, 12), device="cuda", generator=gen), 102
        images = torch.empty(b, 3, 384, 384, device="cuda").uniform_(-1, 1, generator=gen)
        attack.attack_batch(images, ids, (ids != 0).long(), torch.zeros_like(ids, dtype=torch.bool))
        torch.cuda.synchronize()
        print(json.dumps(dict(phase="after_attack", reserved_gb=round(torch.cuda.memory_reserved() / 2 ** 30, 1),
                              allocated_gb=round(torch.cuda.memory_allocated() / 2 ** 30, 1))), flush=True)
        report("after_attack")
    # bench.py's order: the HBM-bound step microbench at batch 256 right before the attention microbench
    from vqattack_amd import ops
    shape = (256, 3, 384, 384)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    g = torch.randn(shape, device="cuda", generator=gen)
    bufs = [x0.clone(), torch.empty_like(x0)]
    for i in range(44):
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
    report("after_step_microbench")


if __name__ == "__main__":
    main()
