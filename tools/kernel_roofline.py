"""Achieved algorithmic GB/s of every C-ABI kernel at BASELINE sizes (warm, back-to-back launches, HIP events).

    python tools/kernel_roofline.py [--batch 64]          -> one JSON line per kernel

Peak: 8 TB/s HBM3E spec; ~6.3 TB/s is what a float4 copy reaches on this chip (MI355X_MICROARCH.md).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd import _hip, ops  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps=20):
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


def report(name, nbytes, ms, note=""):
    gbs = nbytes / ms / 1e6
    print(json.dumps(dict(kernel=name, algorithmic_MB=round(nbytes / 1e6, 1), us=round(ms * 1e3, 1), GBs=round(gbs),
                          frac_of_8TBs=round(gbs / PEAK, 3), note=note)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sections", default="linf,copy,norm,cos,ce,text,block",
                    help="comma list of: linf, copy, norm, cos, ce, text, block")
    args = ap.parse_args()
    sections = set(args.sections.split(","))
    b = args.batch
    shape = (b, 3, 384, 384)
    n = b * 3 * 384 * 384
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    eta = torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen)
    x = torch.clamp(x0 + eta, -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    out = torch.empty_like(x)
    if "copy" in sections:
        copy_probes(x, x0, g, out, n)
    if "linf" in sections:
        linf_section(x, x0, g, eta, out, n)
    if "norm" in sections:
        norm_section(x, x0, g, out, n)
    del eta, out
    if "cos" in sections:
        cos_section(b)
    if "ce" in sections:
        ce_section(b)
    if "text" in sections:
        text_section(b)
    if "block" in sections:
        block_section(b)


def copy_probes(x, x0, g, out, n):
    """What plain copies reach at the SAME footprint, in the same harness (events around back-to-back launches): the
    yardstick for 'the platform's streaming rate' next to the 8 TB/s spec."""
    report("probe: torch copy_ (ATen / hipMemcpyAsync D2D), 1r:1w", 8 * n, timeit(lambda: out.copy_(x)))
    report("probe: stream4 skeleton 1r:1w (vqa_clip_eta_linf, eps=inf -> copy)", 8 * n,
           timeit(lambda: ops.clip_eta_linf(x, float("inf"))), "allocates its output per call")
    report("probe: stream4 skeleton 2r:1w (vqa_linf_project)", 12 * n,
           timeit(lambda: ops.linf_project(x, x0, 0.125, -1, 1, out=out)))
    big = torch.cat([x.reshape(-1), x0.reshape(-1)])          # 2 images' worth: same bytes per launch as the 3r:1w step
    dst = torch.empty_like(big)
    report("probe: torch copy_ at the step kernel's footprint (16 B/el of the image batch)", 16 * n,
           timeit(lambda: dst.copy_(big)))
    del big, dst


def linf_section(x, x0, g, eta, out, n):
    for nt in (1, 3, 5, 7, 13):                 # knob sweeps need the tuning build (VQA_TUNING_LIB=1); skipped otherwise
        if not _hip.set_option(1, nt):
            break
        report("vqa_linf_step [nt mask {}: grad loads {}, stores {}, x/x0 loads {}]".format(
            nt, "nt" if nt & 1 else "plain", "nt" if nt & 2 else ("nt above 256 MB" if nt & 8 else "plain"),
            "nt" if nt & 4 else "plain"), 16 * n,
            timeit(lambda: ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out)), "A/B knob sweep")
    report("vqa_linf_step", 16 * n, timeit(lambda: ops.linf_step(x, g, x0, 0.01, 0.125, -1, 1, out=out)))
    def eager_chain():
        # the reference's op chain between two model calls, as PyTorch-ROCm runs it (SURVEY.md section 2.3, rows 1-13
        # without the host sync): range flags, clone, sign, scale, add, clamp, sub, clamp, add, clamp
        torch.all(torch.ge(x, -1.0)), torch.all(torch.le(x, 1.0))
        xc = x.clone()
        a = torch.clamp(xc + 0.01 * torch.sign(g), -1, 1)
        eta = torch.clamp(a - x0, -0.125, 0.125)
        return torch.clamp(x0 + eta, -1, 1)
    report("reference eager chain on the GPU (13 ATen ops), same 16 B/el basis", 16 * n, timeit(eager_chain),
           "what vqa_linf_step replaces; moves ~92 B/element")
    report("vqa_linf_fgm", 12 * n, timeit(lambda: ops.linf_fgm(x, g, 0.01, -1, 1, out=out)))
    report("vqa_linf_init (eta)", 12 * n, timeit(lambda: ops.linf_init(x0, eta, 0.125, -1, 1, out=out)))
    report("vqa_linf_init (zero)", 8 * n, timeit(lambda: ops.linf_init(x0, None, 0.125, -1, 1, out=out)))
    report("vqa_linf_project", 12 * n, timeit(lambda: ops.linf_project(x, x0, 0.125, -1, 1, out=out)))


def norm_section(x, x0, g, out, n):
    report("vqa_sumsq_per_sample", 4 * n, timeit(lambda: ops.sumsq_per_sample(g)), "two-stage deterministic")
    report("vqa_sumsq_per_sample (diff)", 8 * n, timeit(lambda: ops.sumsq_per_sample(x, sub=x0)))
    report("l2_fgm (sumsq + update)", 4 * n + 12 * n, timeit(lambda: ops.l2_fgm(x, g, 0.5, -1, 1, out=out)),
           "gradient read twice (norm, then update)")
    report("l2_project (sumsq + update)", 8 * n + 12 * n, timeit(lambda: ops.l2_project(x, x0, 2.0, -1, 1, out=out)))
    report("l1_fgm (absmax/ties + update)", 4 * n + 12 * n, timeit(lambda: ops.l1_fgm(x, g, 0.5, -1, 1, out=out)))


def cos_section(b):
    # cosine loss at the VLMO-base per-layer shape; the gradient buffer comes from a Workspace (as in the attack loop)
    ws = ops.Workspace()
    slot = torch.zeros(1, device="cuda")
    rows = b * 617
    a = torch.randn(b, 617, 768, device="cuda")
    t = torch.randn(b, 617, 768, device="cuda")
    report("vqa_neg_cos_rows (loss+grad, D=768)", 12 * rows * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False, ws=ws)), "loss folded in the launch")
    report("vqa_neg_cos_rows (loss only, D=768)", 8 * rows * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False, want_grad=False)))
    report("vqa_neg_cos_rows (loss+grad, D=768), partials only (no in-kernel fold)", 12 * rows * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, None, accumulate=False, ws=ws)), "A/B: cost of the arrival counters")
    for per_cu, inflight in ((0, 1), (8, 2), (8, 1), (4, 2), (0, 2)):     # last = defaults
        if not (_hip.set_option(6, per_cu) and _hip.set_option(7, inflight)):
            break
        report("vqa_neg_cos_rows (loss+grad, D=768) [grid {}, {} row(s) in flight]".format(
            "resident" if per_cu == 0 else "{}/CU".format(per_cu), inflight), 12 * rows * 768,
            timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False, ws=ws)), "A/B knob sweep")
    w = torch.ones(b, 617, dtype=torch.uint8, device="cuda")
    w[:, 0] = 2
    w[:, 10:40] = 0                        # 10 real text tokens of 40, as in the bench workload
    live = int((w != 0).sum())
    report("vqa_neg_cos_rows (loss+grad, D=768, row weights: 30 of 617 rows padded)", 12 * live * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False, row_weight=w, weight_period=b, ws=ws)),
           "bytes of live rows only")
    del a, t
    n_layers = 13 if b <= 64 else 4        # 13 maps at batch 64 = 4.5 GB per launch x 3 buffers
    al = [torch.randn(b, 617, 768, device="cuda") for _ in range(n_layers)]
    tl = [torch.randn(b, 617, 768, device="cuda") for _ in range(n_layers)]
    report("vqa_neg_cos_rows_multi ({} layers, loss+grad, D=768)".format(n_layers), 12 * rows * 768 * n_layers,
           timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, ws=ws), reps=10),
           "the attack loop's launch: every per-layer map of a modality at once")
    report("vqa_neg_cos_rows_multi ({} layers, row weights)".format(n_layers), 12 * live * 768 * n_layers,
           timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, row_weight=w, weight_period=b, ws=ws),
                  reps=10))
    for full in (0, 1, 0, 1):                 # option 12: the general row kernel vs the branch-free whole-chunk one
        if not _hip.set_option(12, full):
            break
        report("vqa_neg_cos_rows_multi ({} layers, row weights) [{}]".format(
            n_layers, "neg_cos_rows_full_kernel" if full else "neg_cos_rows_kernel (general)"), 12 * live * 768 * n_layers,
            timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, row_weight=w, weight_period=b, ws=ws),
                   reps=10), "A/B, alternating")
        report("vqa_neg_cos_rows_multi ({} layers, no weights) [{}]".format(
            n_layers, "neg_cos_rows_full_kernel" if full else "neg_cos_rows_kernel (general)"), 12 * rows * 768 * n_layers,
            timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, ws=ws), reps=10), "A/B, alternating")
    for nt in (0, 5, 6, 7, 4):
        if not _hip.set_option(8, nt):
            break
        report("vqa_neg_cos_rows_multi ({} layers, D=768) [nt mask {}: a loads {}, grad stores {}, b loads {}]".format(
            n_layers, nt, "nt" if nt & 1 else "plain", "nt" if nt & 2 else "plain", "nt" if nt & 4 else "plain"),
            12 * rows * 768 * n_layers,
            timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, ws=ws), reps=10), "A/B knob sweep")
    _hip.set_option(8, -1)                    # back to the shipped rule (nt mask by launch size)
    for per_cu, inflight in ((0, 1), (8, 2), (8, 1), (0, 2)):
        if not (_hip.set_option(6, per_cu) and _hip.set_option(7, inflight)):
            break
        report("vqa_neg_cos_rows_multi ({} layers, D=768) [grid {}, {} row(s) in flight]".format(
            n_layers, "resident" if per_cu == 0 else "{}/CU".format(per_cu), inflight), 12 * rows * 768 * n_layers,
            timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, ws=ws), reps=10), "A/B knob sweep")
    del al, tl, ws
    ws = ops.Workspace()
    n_layers = 25 if b <= 64 else 4
    al = [torch.randn(b, 617, 1024, device="cuda") for _ in range(n_layers)]
    tl = [torch.randn(b, 617, 1024, device="cuda") for _ in range(n_layers)]
    report("vqa_neg_cos_rows (loss+grad, D=1024)", 12 * rows * 1024,
           timeit(lambda: ops.neg_cos_rows(al[0], tl[0], slot, accumulate=False, ws=ws)))
    report("vqa_neg_cos_rows_multi ({} layers, loss+grad, D=1024)".format(n_layers), 12 * rows * 1024 * n_layers,
           timeit(lambda: ops.neg_cos_rows_multi(al, tl, slot, accumulate=False, ws=ws), reps=6),
           "VLMO-large: 25 maps in one launch")


def ce_section(b):
    # MLM cross entropy (B*40 rows of 30522)
    import torch.nn.functional as F
    slot = torch.zeros(1, device="cuda")
    rows = b * 40
    logits = torch.randn(rows, 30522, device="cuda")
    labels = torch.randint(0, 30522, (1, rows), device="cuda")
    labels3 = torch.randint(0, 30522, (3, rows), device="cuda")
    ws = ops.Workspace()
    tuning = _hip.set_option(4, 512)
    for variant, threads in (((2, 256), (2, 512), (3, 512)) if tuning else ((2, 512),)):
        if tuning:
            _hip.set_option(4, threads), _hip.set_option(5, variant)
        tag = "{} threads, {} logits loads".format(threads, "nt" if variant == 3 else "plain")
        report("vqa_ce_rows (loss+grad, K=1) [{}]".format(tag), 8 * rows * 30522,
               timeit(lambda: ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, ws=ws)))
        report("vqa_ce_rows (loss+grad, K=1) [{}], row losses only (no in-kernel fold)".format(tag), 8 * rows * 30522,
               timeit(lambda: ops.mlm_cross_entropy(logits, labels, None, accumulate=False, ws=ws)), "A/B")
    if tuning:
        _hip.set_option(4, 512), _hip.set_option(5, 2)      # defaults
    report("vqa_ce_rows (loss+grad, K=3)", 8 * rows * 30522,
           timeit(lambda: ops.mlm_cross_entropy(logits, labels3, slot, accumulate=False, ws=ws)))
    report("vqa_ce_rows (loss only, K=1)", 4 * rows * 30522,
           timeit(lambda: ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, want_grad=False, ws=ws)))

    # the reference's workload: labels are ignore_index except at the [MASK]-ed answer pieces (adv_attack.py:433-558).
    # Dense (B x 40 rows) with dead rows skipped in the kernel: a dead row costs its zero gradient store, 4 V bytes.
    for dead_frac, live_per_sample in ((0.9, 4), (0.975, 1)):
        lab = torch.full((1, rows), -100, dtype=torch.long, device="cuda")
        for s in range(b):
            lab[0, s * 40 + 1:s * 40 + 1 + live_per_sample] = torch.randint(0, 30522, (live_per_sample,), device="cuda")
        live = int((lab != -100).sum())
        report("vqa_ce_rows (loss+grad, K=1, dense {} x 30522, {:.1%} dead rows, fresh gradient buffer per call)".format(
            rows, 1 - live / rows), (8 * live + 4 * (rows - live)) * 30522,
            timeit(lambda: ops.mlm_cross_entropy(logits, lab, slot, accumulate=False, ws=None, rows_per_sample=40)),
            "no workspace: 8 V bytes per live row + 4 V per dead row (its zero gradient is stored); allocation included")
        ws3 = ops.Workspace()
        report("vqa_ce_rows (loss+grad, K=1, dense {} x 30522, {:.1%} dead rows, attack workspace + row_state)".format(
            rows, 1 - live / rows), 8 * live * 30522,
            timeit(lambda: ops.mlm_cross_entropy(logits, lab, slot, accumulate=False, ws=ws3, rows_per_sample=40)),
            "the external dense model_fn case: gradient buffer kept across iterations, dead rows never stored")
        del ws3
        # live-rows form: what the bundled adapters hand over -- only the live rows exist at all
        small = logits[:b * live_per_sample].contiguous().reshape(b, live_per_sample, 30522)
        lab_s = torch.randint(0, 30522, (1, b * live_per_sample), device="cuda")
        ws2 = ops.Workspace()
        report("vqa_ce_rows (loss+grad, K=1, live-rows form: {} x 30522)".format(b * live_per_sample),
               8 * b * live_per_sample * 30522,
               timeit(lambda: ops.mlm_cross_entropy(small, lab_s, slot, accumulate=False, ws=ws2,
                                                    rows_per_sample=live_per_sample)),
               "latency-bound: {} workgroups on 256 CUs".format(b * live_per_sample))

    def torch_ce():
        lg = logits.detach().requires_grad_(True)
        F.cross_entropy(lg, labels[0]).backward()
    report("torch F.cross_entropy fwd+bwd (K=1), same bytes basis", 8 * rows * 30522, timeit(torch_ce),
           "PyTorch-ROCm eager reference for the op the kernel replaces")


def block_section(b):
    """The white box's block glue (csrc/block.hip) at the VLMO-base bench layout: S = 14 text + 577 image tokens, D = 768,
    hidden 3072.  Algorithmic bytes = every operand once."""
    for nt in (0, 1, 3):                      # tuning build: A/B of the non-temporal hints; shipped build: one pass
        tuned = _hip.set_option(10, nt)
        if tuned or nt == 3:
            _block_pass(b, " [nt mask {}]".format(nt) if tuned else "")


def _block_pass(b, tag):
    _report = globals()["report"]

    def report(name, *a, **kw):
        _report(name + tag, *a, **kw)
    s, t, d = 591, 14, 768
    rows = b * s
    dev = "cuda"
    x, r = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
    gam, bet, sc = torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.ones(d, device=dev)
    y, x_out = torch.empty(rows, d, device=dev), torch.empty(rows, d, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    e = 4 * rows * d
    report("vqa_ln_fwd (LayerNorm only)", 2 * e, timeit(lambda: ops.ln_fwd(x, gam, bet, y, mean, rstd, 1e-6)))
    report("vqa_ln_fwd (residual add of the previous branch + LayerNorm)", 4 * e,
           timeit(lambda: ops.ln_fwd(x, gam, bet, y, mean, rstd, 1e-6, r0=r, rscale=sc, x_out=x_out)))
    y0, y1 = torch.empty(b * t, d, device=dev), torch.empty(b * (s - t), d, device=dev)
    report("vqa_ln_fwd (residual add + text / image LayerNorm, split output)", 4 * e,
           timeit(lambda: ops.ln_fwd(x, gam, bet, y0, mean, rstd, 1e-6, r0=r, rscale=sc, x_out=x_out, gamma1=gam,
                                     beta1=bet, y1=y1, period=s, split=t)))
    ops.ln_fwd(x, gam, bet, y, mean, rstd, 1e-6)
    dy, g_a, g_inj = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
    dx, dr = torch.empty(rows, d, device=dev), torch.empty(rows, d, device=dev)
    report("vqa_ln_bwd (LN backward + residual gradient + layer-scaled branch gradient)", 5 * e,
           timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gam, dx, g_a=g_a, rscale=sc, dr0=dr)))
    dr0, dr1 = torch.empty(b * t, d, device=dev), torch.empty(b * (s - t), d, device=dev)
    report("vqa_ln_bwd (+ the loss gradient of the feature map, split branch gradient)", 6 * e,
           timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gam, dx, g_a=g_a, g_inj=g_inj, rscale=sc, dr0=dr0, dr1=dr1,
                                     period=s, split=t)))

    def eager_stage():
        # what autograd's eager ops run for the same stage: addcmul, layer_norm, two slice copies
        x1 = torch.addcmul(x.view(b, s, d), sc, r.view(b, s, d))
        return torch.nn.functional.layer_norm(x1[:, :t].contiguous(), (d,), gam, bet, 1e-6), \
            torch.nn.functional.layer_norm(x1[:, t:].contiguous(), (d,), gam, bet, 1e-6)
    report("eager ATen chain for the split forward stage (addcmul + slice copies + layer_norm), same 16 B/el basis",
           4 * e, timeit(eager_stage), "what vqa_ln_fwd replaces")
    del x, r, y, x_out, dy, g_a, g_inj, dx, dr, y0, y1, dr0, dr1
    n = b * 577 * 3072                       # the image expert's hidden activations
    h, da = torch.randn(n, device=dev), torch.randn(n, device=dev)
    a = torch.empty_like(h)
    report("vqa_gelu_fwd (image expert hidden, {} x 3072)".format(b * 577), 8 * n, timeit(lambda: ops.gelu_fwd(h, out=a)))
    report("vqa_gelu_bwd (in place on da)", 12 * n, timeit(lambda: ops.gelu_bwd(h, da)))
    report("torch F.gelu, same bytes", 8 * n, timeit(lambda: torch.nn.functional.gelu(h)), "allocates its output")


def text_section(b):
    # text gradient gather + candidate scoring
    tg = torch.randn(b, 40, 768, device="cuda")
    idx = list(range(40))
    report("vqa_gather_rows (B,40,768)[:, 0..39]", 8 * b * 40 * 768, timeit(lambda: ops.gather_rows(tg, idx)),
           "launch-bound at this size (host index check dominates)")
    word = torch.randn(30522, 768, device="cuda") * 0.02
    pos = torch.randn(512, 768, device="cuda") * 0.02
    typ = torch.randn(2, 768, device="cuda") * 0.02
    gamma, beta = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
    e_ori = torch.randn(b, 40, 768, device="cuda")
    ncand = b * 8 * 5
    cand = torch.stack([torch.arange(ncand, device="cuda") // 40, torch.randint(1, 9, (ncand,), device="cuda"),
                        torch.randint(1, 9, (ncand,), device="cuda"), torch.randint(1000, 30522, (ncand,), device="cuda")],
                       dim=1).to(torch.int32)
    report("vqa_cand_dir_sim ({} candidates)".format(ncand), ncand * 768 * 4 * 5,
           timeit(lambda: ops.cand_dir_sim(word, pos, typ, gamma, beta, 1e-12, e_ori, tg, cand)), "latency-bound")


if __name__ == "__main__":
    main()
