"""Which shader clock does the card hold under each kernel?  Loops one workload for a few seconds while polling
`rocm-smi --showclocks` (sclk) and `--showpower` from a helper thread.  usage: python tools/clock_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import attention  # noqa: E402


def poll(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
            p = re.search(r"Power \(W\): ([\d.]+)", t)
            out.append((int(m.group(1)) if m else None, float(p.group(1)) if p else None))
        except Exception as e:  # noqa: BLE001
            out.append((None, str(e)[:40]))
        time.sleep(0.3)


def run(name, fn, seconds=4.0):
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    print(name, "ms/call %.3f" % (dt / n * 1e3), "samples (sclk MHz, W):", out[1:-1], flush=True)


def main():
    B, H, S, D = 64, 12, 587, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, H, D, device="cuda", generator=g)
    spad = (S + 31) // 32 * 32
    store = torch.zeros(1, H, S, spad, device="cuda")
    store[..., :S] = torch.randn(1, H, S, S, device="cuda", generator=g) * 0.02
    bias = store[..., :S].expand(B, -1, -1, -1)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    run("idle", lambda: None, 1.5)
    run("hip attention fwd", lambda: attention.attention_forward(q, k, v, bias))
    o, lse = attention.attention_forward(q, k, v, bias)
    go = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    run("hip attention bwd", lambda: attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1],
                                                         dqkv[:, :, 2], D ** -0.5))
    qt, kt, vt = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    run("sdpa fwd", lambda: F.scaled_dot_product_attention(qt, kt, vt, attn_mask=bias))
    a = torch.randn(36928, 768, device="cuda")
    w = torch.randn(3072, 768, device="cuda")
    run("sgemm 36928x768x3072", lambda: F.linear(a, w))


if __name__ == "__main__":
    main()
