"""A handful of launches of ONE hot kernel at a BASELINE size, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
SEPARATE runs, kernel-trace only -- see /opt/skills/guides/MI355X_MICROARCH.md, section HBM).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o <phase>_fetch --output-format csv -- python3 tools/pmc_step.py <phase>

phases: step64 | step256 (fused L-inf step, batch 64 / 256), cos13 (vqa_neg_cos_rows_multi: 13 VLMO-base maps of batch 64
in one launch, 10 real text tokens of 40 like the bench workload), cos25 (25 VLMO-large maps, D = 1024), ce (MLM cross
entropy, 2560 x 30522), sumsq (per-sample sum of squares, batch 64).  One phase per run keeps the per-launch counters of
different shapes apart (the B=64 and B=256 step launches share a grid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import ops  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "step64"
slot = torch.zeros(1, device="cuda")
if phase in ("step64", "step256"):
    batch = 64 if phase == "step64" else 256
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out)
elif phase in ("cos13", "cos25"):
    n_layers, d = (13, 768) if phase == "cos13" else (25, 1024)
    ws = ops.Workspace()
    al = [torch.randn(64, 617, d, device="cuda") for _ in range(n_layers)]
    tl = [torch.randn(64, 617, d, device="cuda") for _ in range(n_layers)]
    w = torch.ones(64, 617, dtype=torch.uint8, device="cuda")
    w[:, 0] = 2
    w[:, 10:40] = 0
    for _ in range(3):
        ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, row_weight=w, weight_period=64, ws=ws)
elif phase == "ce":
    ws = ops.Workspace()
    logits = torch.randn(64 * 40, 30522, device="cuda")
    labels = torch.randint(0, 30522, (1, 64 * 40), device="cuda")
    for _ in range(3):
        ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, ws=ws)
elif phase == "sumsq":
    g = torch.randn(64, 3, 384, 384, device="cuda")
    for _ in range(3):
        ops.sumsq_per_sample(g)
else:
    raise SystemExit("unknown phase " + phase)
torch.cuda.synchronize()
