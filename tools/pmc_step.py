"""A handful of launches of ONE hot kernel at a BASELINE size, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
SEPARATE runs, kernel-trace only -- see /opt/skills/guides/MI355X_MICROARCH.md, section HBM).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o <phase>_fetch --output-format csv -- python3 tools/pmc_step.py <phase>

phases: step64 | step256 (fused L-inf step, batch 64 / 256), cos13 (vqa_neg_cos_rows_multi: 13 VLMO-base maps of batch 64
in one launch, in the bench workload's token layout: its seeded questions of 6..14 real tokens, padding trimmed), cos25
(25 VLMO-large maps, D = 1024), ce (MLM cross entropy, 2560 x 30522), ce_dead (the same launch with the reference's
label pattern, 1 live row of 40, from the attack's workspace: dead rows are not stored), ce_dead_fresh (the same without
a workspace: a fresh gradient buffer whose dead rows are zero-filled by the launch), attn (the white box's attention
forward + backward at the bench shape, saved-scores / dS^T-workspace form), sumsq (per-sample sum of squares, batch 64).
One phase per run keeps the per-launch counters of different shapes apart (the B=64 and B=256 step launches share a
grid).  The phase prints `PMC_SHAPE <json>`: the launch shape, stored with the counters (tools/pmc_summary.py --json) so
that bench.py copies a recorded figure only into a line whose launch has exactly that shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqattack_amd import ops  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "step64"
slot = torch.zeros(1, device="cuda")
shape_record = None


def bench_text_layout(batch=64, text_len=40):
    """The default bench run's question batch (bench.synthetic_questions, rank 0): (trimmed text length, row weights)."""
    import bench
    _, masks, n_words = bench.synthetic_questions(batch, text_len, seed=100, device="cuda")
    tlen = max(n_words) + 2
    w = torch.cat([masks[:, :tlen].to(torch.uint8), torch.ones(batch, 577, dtype=torch.uint8, device="cuda")], dim=1)
    w[:, 0] = 2
    return tlen, w.contiguous()


if phase in ("step64", "step256"):
    batch = 64 if phase == "step64" else 256
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    out = torch.empty_like(x)
    shape_record = dict(op="linf_step", elements=x.numel())
    for _ in range(3):
        ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out)
elif phase in ("cos13", "cos25"):
    n_layers, d = (13, 768) if phase == "cos13" else (25, 1024)
    ws = ops.Workspace()
    tlen, w = bench_text_layout()
    tokens = tlen + 577
    al = [torch.randn(64, tokens, d, device="cuda") for _ in range(n_layers)]
    tl = [torch.randn(64, tokens, d, device="cuda") for _ in range(n_layers)]
    shape_record = dict(op="neg_cos_rows_multi", maps=n_layers, batch=64, tokens=tokens, dim=d,
                        live_rows=int((w != 0).sum()))
    for _ in range(3):
        ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, row_weight=w, weight_period=64, ws=ws)
elif phase == "ce":
    ws = ops.Workspace()
    logits = torch.randn(64 * 40, 30522, device="cuda")
    labels = torch.randint(0, 30522, (1, 64 * 40), device="cuda")
    shape_record = dict(op="mlm_cross_entropy", rows=64 * 40, vocab=30522, live_rows=64 * 40)
    for _ in range(3):
        ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, ws=ws)
elif phase in ("ce_dead", "ce_dead_fresh"):
    # the reference's label pattern: 1 live position of 40 per sample (97.5 % dead rows): a dead row's logits are never
    # fetched.  With the attack's workspace its gradient row is not stored either (zero-filled once at allocation):
    # expected traffic ~ 8 V x 64 bytes; without a workspace the zero rows are written: + 4 V x 2496 bytes
    ws = ops.Workspace() if phase == "ce_dead" else None
    logits = torch.randn(64 * 40, 30522, device="cuda")
    labels = torch.full((1, 64 * 40), -100, dtype=torch.long, device="cuda")
    labels[0, 4::40] = torch.randint(0, 30522, (64,), device="cuda")
    shape_record = dict(op="mlm_cross_entropy", rows=64 * 40, vocab=30522, live_rows=64, workspace=ws is not None)
    for _ in range(3):
        ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, ws=ws, rows_per_sample=40)
elif phase == "attn":
    # the white box's attention at the bench shape, in the form the attack's autograd path runs: a forward that saves
    # its scores, a backward that starts from them and keeps dS^T in the transient workspace (4 products)
    from vqattack_amd import attention
    b, h, sq = 64, 12, bench_text_layout()[0] + 577
    shape_record = dict(op="attention", batch=b, heads=h, seq=sq, head_dim=64, bias=True)
    gen = torch.Generator(device="cuda").manual_seed(2)
    qkv = torch.randn(b, sq, 3, h, 64, device="cuda", generator=gen)
    store = torch.zeros(1, h, sq, (sq + 31) // 32 * 32, device="cuda")
    store[..., :sq] = torch.randn(1, h, sq, sq, device="cuda", generator=gen) * 0.02
    bias = store[..., :sq].expand(b, -1, -1, -1)
    bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    go = torch.randn(b, sq, h, 64, device="cuda", generator=gen)
    dqkv = torch.empty_like(qkv)
    for _ in range(3):
        o, lse, scores = attention._forward(q, k, v, bias, bstr, 0.125, save_scores=True)
        assert scores is not None
        attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], 0.125,
                            scores=scores)
elif phase in ("ln_fwd", "ln_bwd", "gelu"):
    # the white box's block glue (csrc/block.hip) at the bench layout: 64 x 591 tokens x 768, image-expert hidden 3072
    b, s, d = 64, bench_text_layout()[0] + 577, 768
    t, rows = s - 577, 64 * (bench_text_layout()[0] + 577)
    dev = "cuda"
    if phase == "gelu":
        n = b * 577 * 3072
        h, da = torch.randn(n, device=dev), torch.randn(n, device=dev)
        a = torch.empty_like(h)
        shape_record = dict(op="gelu_fwd+gelu_bwd", elements=n)
        for _ in range(3):
            ops.gelu_fwd(h, out=a)
            ops.gelu_bwd(h, da)
    else:
        x, r = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
        gam, bet, sc = torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.ones(d, device=dev)
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        x_out = torch.empty(rows, d, device=dev)
        y0, y1 = torch.empty(b * t, d, device=dev), torch.empty(b * (s - t), d, device=dev)
        ops.ln_fwd(x, gam, bet, y0, mean, rstd, 1e-6, r0=r, rscale=sc, x_out=x_out, gamma1=gam, beta1=bet, y1=y1,
                   period=s, split=t)
        if phase == "ln_fwd":     # residual add + text / image LayerNorm + split: read x, r; write x_out, y = 16 B/element
            shape_record = dict(op="ln_fwd", rows=rows, dim=d, residual=True, split=True)
            for _ in range(3):
                ops.ln_fwd(x, gam, bet, y0, mean, rstd, 1e-6, r0=r, rscale=sc, x_out=x_out, gamma1=gam, beta1=bet, y1=y1,
                           period=s, split=t)
        else:                     # LN backward + residual + loss gradient + split branch gradient: 24 B/element
            dy0, dy1 = torch.randn_like(y0), torch.randn_like(y1)
            g_a, g_inj = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
            dx = torch.empty(rows, d, device=dev)
            dr0, dr1 = torch.empty_like(y0), torch.empty_like(y1)
            shape_record = dict(op="ln_bwd", rows=rows, dim=d, residual=True, injected=True, split=True)
            for _ in range(3):
                ops.ln_bwd(dy0, x_out, mean, rstd, gam, dx, dy1=dy1, gamma1=gam, g_a=g_a, g_inj=g_inj, rscale=sc, dr0=dr0,
                           dr1=dr1, period=s, split=t)
elif phase == "sumsq":
    g = torch.randn(64, 3, 384, 384, device="cuda")
    shape_record = dict(op="sumsq_per_sample", batch=64, elements=g.numel())
    for _ in range(3):
        ops.sumsq_per_sample(g)
else:
    raise SystemExit("unknown phase " + phase)
torch.cuda.synchronize()
print("PMC_SHAPE " + json.dumps(shape_record, sort_keys=True), flush=True)
