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
from vqattack_amd import ops  # noqa: E402

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
    args = ap.parse_args()
    b = args.batch
    shape = (b, 3, 384, 384)
    n = b * 3 * 384 * 384
    gen = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    eta = torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen)
    x = torch.clamp(x0 + eta, -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    out = torch.empty_like(x)
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
    report("vqa_sumsq_per_sample", 4 * n, timeit(lambda: ops.sumsq_per_sample(g)), "two-stage deterministic")
    report("vqa_sumsq_per_sample (diff)", 8 * n, timeit(lambda: ops.sumsq_per_sample(x, sub=x0)))
    report("l2_fgm (sumsq + update)", 4 * n + 12 * n, timeit(lambda: ops.l2_fgm(x, g, 0.5, -1, 1, out=out)),
           "gradient read twice (norm, then update)")
    report("l2_project (sumsq + update)", 8 * n + 12 * n, timeit(lambda: ops.l2_project(x, x0, 2.0, -1, 1, out=out)))
    report("l1_fgm (absmax/ties + update)", 4 * n + 12 * n, timeit(lambda: ops.l1_fgm(x, g, 0.5, -1, 1, out=out)))
    del eta, out
    # cosine loss at the VLMO-base per-layer shape
    a = torch.randn(b, 617, 768, device="cuda")
    t = torch.randn(b, 617, 768, device="cuda")
    slot = torch.zeros(1, device="cuda")
    rows = b * 617
    report("vqa_neg_cos_rows (loss+grad, D=768)", 12 * rows * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False)), "incl. sum_partials + grad alloc")
    report("vqa_neg_cos_rows (loss only, D=768)", 8 * rows * 768,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False, want_grad=False)))
    del a, t
    a = torch.randn(b, 617, 1024, device="cuda")
    t = torch.randn(b, 617, 1024, device="cuda")
    report("vqa_neg_cos_rows (loss+grad, D=1024)", 12 * rows * 1024,
           timeit(lambda: ops.neg_cos_rows(a, t, slot, accumulate=False)))
    del a, t
    # MLM cross entropy (B*40 rows of 30522)
    rows = b * 40
    logits = torch.randn(rows, 30522, device="cuda")
    labels = torch.randint(0, 30522, (1, rows), device="cuda")
    report("vqa_ce_rows (loss+grad, K=1)", 8 * rows * 30522,
           timeit(lambda: ops.mlm_cross_entropy(logits, labels, slot, accumulate=False)), "row re-read from L2 not counted")
    labels3 = torch.randint(0, 30522, (3, rows), device="cuda")
    report("vqa_ce_rows (loss+grad, K=3)", 8 * rows * 30522,
           timeit(lambda: ops.mlm_cross_entropy(logits, labels3, slot, accumulate=False)))
    import torch.nn.functional as F

    def torch_ce():
        lg = logits.detach().requires_grad_(True)
        F.cross_entropy(lg, labels[0]).backward()
    report("torch F.cross_entropy fwd+bwd (K=1), same bytes basis", 8 * rows * 30522, timeit(torch_ce),
           "PyTorch-ROCm eager reference for the op the kernel replaces")
    del logits
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
