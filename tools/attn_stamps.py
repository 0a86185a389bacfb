"""Where a tile of the attention loops spends its cycles: s_memtime stamps of the TUNING build (csrc/attn.hip, VQA_STAMP).

    python -m vqattack_amd.build --tuning && VQA_TUNING_LIB=1 python tools/attn_stamps.py

One forward (saving scores) and one backward from saved scores at the bench shape with the stamps on; per kernel the
share of each segment in a wave's loop time and the mean cycles per tile.  The stamped build serialises what the real
kernel overlaps (every stamp drains the LDS queue), so only the SHARES are meaningful.
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip, attention  # noqa: E402

B, H, S, D = int(os.environ.get("B", "64")), 12, int(os.environ.get("S", "591")), 64
SEGMENTS = {
    "attn_fwd_kernel": ["prefetch issue", "S chain issue", "softmax", "P.V chains issue",
                        "wait for prefetch + LDS publish", "score stores issue", "barrier"],
    "attn_bwd_dkv_kernel": ["prefetch issue", "LDS reads + P", "dP chain issue", "dS", "dV/dK chains issue",
                            "wait for prefetch + LDS publish", "dS stores issue", "barrier"],
}


def main():
    lib = _hip.lib()
    if not hasattr(lib, "vqa_attn_set_stamps"):
        raise SystemExit("needs the tuning build: python -m vqattack_amd.build --tuning; VQA_TUNING_LIB=1")
    lib.vqa_attn_set_stamps.restype, lib.vqa_attn_set_stamps.argtypes = ctypes.c_int, [ctypes.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, H, D, device="cuda", generator=g)
    spad = (S + 31) // 32 * 32
    store = torch.zeros(1, H, S, spad, device="cuda")
    store[..., :S] = torch.randn(1, H, S, S, device="cuda", generator=g) * 0.02
    bias = store[..., :S].expand(B, -1, -1, -1)
    bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    go = torch.randn(B, S, H, D, device="cuda", generator=g)
    dqkv = torch.empty_like(qkv)

    def run():
        o, lse, scores = attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True)
        attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], D ** -0.5,
                            scores=scores)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = torch.zeros(20, dtype=torch.int64, device="cuda")
    assert lib.vqa_attn_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    assert lib.vqa_attn_set_stamps(None) == 0
    host = buf.cpu().tolist()
    for i, (name, segs) in enumerate(SEGMENTS.items()):
        sums, tiles = host[10 * i:10 * i + len(segs)], host[10 * i + 9]
        total = float(sum(sums))
        print(json.dumps(dict(kernel=name, wave_tiles=tiles, stamp_ticks_per_tile=round(total / max(tiles, 1), 1),
                              shares={s: round(x / total, 4) for s, x in zip(segs, sums)},
                              ticks_per_tile={s: round(x / max(tiles, 1), 1) for s, x in zip(segs, sums)})), flush=True)


if __name__ == "__main__":
    main()
