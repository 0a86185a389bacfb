"""A/B of two builds of the kernel library on the attention calls of the bench shape, alternating on one box.

    python tools/attn_ab.py vqattack_amd/lib/libvqattack_hip_prev.so vqattack_amd/lib/libvqattack_hip.so [rounds]

Per build and round: the forward that saves its scores, and the backward from saved scores (delta + dK/dV + dQ), each
timed with HIP events over 20 back-to-back launches after 5 warm-up launches; then the largest absolute difference of
the two builds' outputs and gradients (same inputs).  S=591 (VLMO-base, 384 px) unless S is set in the environment.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip, attention  # noqa: E402

B, H, S, D = int(os.environ.get("B", "64")), 12, int(os.environ.get("S", "591")), 64


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
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
    paths = [a for a in sys.argv[1:] if not a.isdigit()][:2]       # one path: that build alone (for a counter pass)
    rounds = int(next((a for a in sys.argv[1:] if a.isdigit()), "3"))
    libs = [_hip.load_library(os.path.abspath(p)) for p in paths]
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, H, D, device="cuda", generator=g)
    spad = (S + 31) // 32 * 32
    store = torch.zeros(1, H, S, spad, device="cuda")
    store[..., :S] = torch.randn(1, H, S, S, device="cuda", generator=g) * 0.02
    bias = store[..., :S].expand(B, -1, -1, -1)
    bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    go = torch.randn(B, S, H, D, device="cuda", generator=g)
    flops = 4.0 * B * H * S * S * D
    results = []
    for which, handle in enumerate(libs):
        _hip._lib = handle
        o, lse, scores = attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True)
        dqkv = torch.empty_like(qkv)
        attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], D ** -0.5,
                            scores=scores)
        results.append((o.clone(), lse.clone(), dqkv.clone()))
    names = ("o", "lse", "dqkv")
    if len(libs) == 2:
        print(json.dumps(dict(what="largest |A - B|", **{n: float((a - b).abs().max()) for n, a, b in
                                                         zip(names, results[0], results[1])},
                              largest_grad=float(results[0][2].abs().max()))), flush=True)
    ablate = int(os.environ.get("ABLATE", "0"))             # tuning build only: timing ablations (wrong results)
    if ablate:
        import ctypes
        for handle in libs:
            handle.vqa_attn_set_ablation.restype, handle.vqa_attn_set_ablation.argtypes = ctypes.c_int, [ctypes.c_int]
            assert handle.vqa_attn_set_ablation(ablate) == 0
        print(json.dumps(dict(ablation_bits=ablate, note="timings only; the results of this run are wrong")), flush=True)
    chunks = [int(c) for c in os.environ.get("CHUNKS", "").split(",") if c]      # tuning build: option 9 per round
    for rnd in range(max(rounds, len(chunks))):
        if chunks:
            import ctypes
            for handle in libs:
                handle.vqa_set_option.restype, handle.vqa_set_option.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int]
                assert handle.vqa_set_option(9, chunks[rnd % len(chunks)]) == 0
            print(json.dumps(dict(backward_batch_chunk=chunks[rnd % len(chunks)])), flush=True)
        for which, handle in enumerate(libs):
            _hip._lib = handle
            fwd = timeit(lambda: attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True))
            o, lse, scores = attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True)
            dqkv = torch.empty_like(qkv)
            bwd = timeit(lambda: attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1],
                                                     dqkv[:, :, 2], D ** -0.5, scores=scores))
            print(json.dumps(dict(round=rnd, lib=os.path.basename(paths[which]), fwd_ms=round(fwd, 4),
                                  bwd_ms=round(bwd, 4), fwd_of_peak=round(flops / fwd / 1e9 / 157.3, 4),
                                  bwd_of_peak=round(2 * flops / bwd / 1e9 / 157.3, 4))), flush=True)


if __name__ == "__main__":
    main()
