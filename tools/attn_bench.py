"""fp32 attention at the bench shape (VLMO-base, B=64, H=12, S=587 after padding trim, d=64; relative-position bias
shared over the batch): the hand-written MFMA kernels (csrc/attn.hip) vs PyTorch-ROCm's scaled_dot_product_attention."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip, attention  # noqa: E402

B, H, S, D = 64, 12, int(os.environ.get("S", "587")), 64


def timeit(fn, reps=10):
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
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, H, D, device="cuda", generator=g)
    spad = (S + 31) // 32 * 32
    store = torch.zeros(1, H, S, spad, device="cuda")
    store[..., :S] = torch.randn(1, H, S, S, device="cuda", generator=g) * 0.02
    bias = store[..., :S].expand(B, -1, -1, -1) if os.environ.get("BIAS", "1") == "1" else None
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    flops_fwd = 4.0 * B * H * S * S * D
    ms = timeit(lambda: attention.attention_forward(q, k, v, bias))
    print(json.dumps(dict(what="hip fwd", ms=round(ms, 3), TFLOPs=round(flops_fwd / ms / 1e9, 1))), flush=True)
    qt, kt, vt = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    ms = timeit(lambda: F.scaled_dot_product_attention(qt, kt, vt, attn_mask=bias))
    print(json.dumps(dict(what="sdpa fwd", ms=round(ms, 3), TFLOPs=round(flops_fwd / ms / 1e9, 1))), flush=True)
    if hasattr(attention, "attention"):
        go = torch.randn(B, S, H, D, device="cuda", generator=g)
        qkv_l = qkv.clone().requires_grad_(True)

        def hip_fb():
            o = attention.attention(qkv_l[:, :, 0], qkv_l[:, :, 1], qkv_l[:, :, 2], bias)
            o.backward(go)
            qkv_l.grad = None
        ms = timeit(hip_fb)
        print(json.dumps(dict(what="hip fwd+bwd", ms=round(ms, 3), TFLOPs=round(3.5 * flops_fwd / ms / 1e9, 1))),
              flush=True)

        def hip_packed():
            o = attention.self_attention_packed(qkv_l, bias)
            o.backward(go)
            qkv_l.grad = None
        ms = timeit(hip_packed)
        print(json.dumps(dict(what="hip packed fwd+bwd", ms=round(ms, 3),
                              TFLOPs=round(3.5 * flops_fwd / ms / 1e9, 1))), flush=True)
        bstr = (bias.stride(0), bias.stride(1), bias.stride(2)) if bias is not None else None
        ms = timeit(lambda: attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True))
        print(json.dumps(dict(what="hip fwd saving its scores", ms=round(ms, 3),
                              TFLOPs=round(flops_fwd / ms / 1e9, 1))), flush=True)
        o, lse, scores = attention._forward(q, k, v, bias, bstr, D ** -0.5, save_scores=True)
        dqkv = torch.empty_like(qkv)
        forms = [("hip bwd kernels, from saved scores", dict(scores=scores), None),
                 ("hip bwd kernels, dS workspace only", dict(), None),
                 ("hip bwd kernels, recompute form", dict(workspace=False), None)]
        for what, kw, knob in forms:
            ms = timeit(lambda: attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1],
                                                    dqkv[:, :, 2], D ** -0.5, **kw))
            print(json.dumps(dict(what=what, ms=round(ms, 3), TFLOPs=round(2.5 * flops_fwd / ms / 1e9, 1))),
                  flush=True)

        def sdpa_fb():
            o = F.scaled_dot_product_attention(qkv_l[:, :, 0].transpose(1, 2), qkv_l[:, :, 1].transpose(1, 2),
                                               qkv_l[:, :, 2].transpose(1, 2), attn_mask=bias)
            o.backward(go.transpose(1, 2))
            qkv_l.grad = None
        ms = timeit(sdpa_fb)
        print(json.dumps(dict(what="sdpa fwd+bwd", ms=round(ms, 3), TFLOPs=round(3.5 * flops_fwd / ms / 1e9, 1))),
              flush=True)


if __name__ == "__main__":
    main()
