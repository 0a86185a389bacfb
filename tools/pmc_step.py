"""A handful of launches of the hot kernels at BASELINE sizes, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
in SEPARATE runs, kernel-trace only -- see /opt/skills/guides/MI355X_MICROARCH.md, section HBM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import ops  # noqa: E402

for batch in (64, 256):
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out)
    torch.cuda.synchronize()
    del x0, x, g, out
# cosine loss + gradient at the VLMO-base batch-64 layer shape (B, T+N, D) = (64, 617, 768)
a = torch.randn(64, 617, 768, device="cuda")
b = torch.randn(64, 617, 768, device="cuda")
slot = torch.zeros(1, device="cuda")
for _ in range(3):
    ops.neg_cos_rows(a, b, slot, accumulate=False)
torch.cuda.synchronize()

# MLM cross entropy at the B=64 shape (2560 rows x 30522) and the per-sample reductions of the L2 path
logits = torch.randn(64 * 40, 30522, device="cuda")
labels = torch.randint(0, 30522, (1, 64 * 40), device="cuda")
for _ in range(3):
    ops.mlm_cross_entropy(logits, labels, slot, accumulate=False)
torch.cuda.synchronize()
del logits
g = torch.randn(64, 3, 384, 384, device="cuda")
for _ in range(3):
    ops.sumsq_per_sample(g)
torch.cuda.synchronize()
