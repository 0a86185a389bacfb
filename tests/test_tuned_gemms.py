"""Recorded library GEMM solutions (``whitebox/tuned_gemms.py``: PyTorch TunableOp, read-only) -- the tracked file loads on
this software stack, tuning stays off, a recorded shape still computes the library's fp32 GEMM, and a file recorded for
another stack is ignored instead of trusted.  Runs in a child of the pre-GPU fork server: TunableOp is process-global
state and must not leak into the other GPU tests."""
import multiprocessing
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(out_path, bad_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    from vqattack_amd.whitebox import tuned_gemms
    lines = []
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    # a shape of the default bench workload: the image expert's fc1 backward, dX = dY (36928 x 3072) @ W (3072 x 768)
    a = torch.randn(36928, 3072, device=dev, generator=g)
    w = torch.randn(3072, 768, device=dev, generator=g) * 0.02
    bias = torch.randn(3072, device=dev, generator=g)
    y = torch.randn(36928, 768, device=dev, generator=g)
    ref_mm = torch.mm(a, w)
    ref_addmm = torch.addmm(bias, y, w.t())
    lines.append("bad_file_accepted {}".format(tuned_gemms.enable(bad_path)))
    ok = tuned_gemms.enable()
    st = tuned_gemms.status()
    lines.append("enabled {} tuning {} entries {}".format(ok, st["tuning"], st["entries"]))
    got_mm, got_addmm = torch.mm(a, w), torch.addmm(bias, y, w.t())
    scale = float(ref_mm.abs().max())
    lines.append("mm_err {:.3e}".format(float((got_mm - ref_mm).abs().max()) / scale))
    lines.append("addmm_err {:.3e}".format(float((got_addmm - ref_addmm).abs().max()) / float(ref_addmm.abs().max())))
    # every recorded entry: the tuned solution is bit-stable run to run (a split-K / atomic solution would not be) and is
    # the library's fp32 product up to the summation order
    worst, unstable, checked = 0.0, [], 0
    for row in open(tuned_gemms.DEFAULT_FILE).read().splitlines():
        op, _, rest = row.partition(",")
        if not op.startswith("Gemm") or "Batched" in op:
            continue
        sig = rest.split(",")[0].split("_")
        m, n, k = int(sig[1]), int(sig[2]), int(sig[3])
        if 4 * m * n > (2 << 30):               # the 20 GB MLM-head product: not worth the box's time here
            continue
        gg = torch.Generator(device=dev).manual_seed(m + n + k)
        if op.endswith("_TN"):                  # addmm(bias (m), y (n, k), w (m, k)^T) -> (n, m): F.linear
            yy = torch.randn(n, k, device=dev, generator=gg)
            ww = torch.randn(m, k, device=dev, generator=gg) * k ** -0.5
            bb = torch.randn(m, device=dev, generator=gg)
            run = (lambda: torch.addmm(bb, yy, ww.t())) if op.startswith("GemmAndBias") else (lambda: torch.mm(yy, ww.t()))
        else:                                   # NN: mm(a (n, k), w (k, m)) -> (n, m): the input-gradient product
            yy = torch.randn(n, k, device=dev, generator=gg)
            ww = torch.randn(k, m, device=dev, generator=gg) * k ** -0.5
            run = lambda: torch.mm(yy, ww)      # noqa: E731
        one, two = run(), run()
        if not torch.equal(one, two):
            unstable.append(rest.split(",")[0])
        torch.cuda.tunable.enable(False)
        ref = run()
        torch.cuda.tunable.enable(True)
        worst = max(worst, float((one - ref).abs().max()) / float(ref.abs().max()))
        checked += 1
        del one, two, ref, yy, ww
    lines.append("entries_checked {}".format(checked))
    lines.append("entries_unstable {}".format(",".join(unstable) or "none"))
    lines.append("entries_worst_err {:.3e}".format(worst))
    # a white-box attack with the recorded solutions active, twice: same bits (images and losses)
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    cfg = vlmo_base(384)
    white = FrozenVlmo(cfg, seed=0).to(dev)
    attack = BatchedVQAttack(VlmoAttackAdapters(white), "vlmo", white.embedding_tables(), AttackConfig(budget=3))
    ids = torch.zeros(64, 40, dtype=torch.long, device=dev)
    ids[:, 0], ids[:, 1:13], ids[:, 13] = 101, torch.randint(1000, 30522, (64, 12), device=dev, generator=g), 102
    images = torch.empty(64, 3, 384, 384, device=dev).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(images).uniform_(-0.125, 0.125, generator=g)
    runs = [attack.attack_batch(images, ids, (ids != 0).long(), torch.zeros_like(ids, dtype=torch.bool), init_eta=eta)
            for _ in range(2)]
    lines.append("attack_bitwise_equal {}".format(torch.equal(runs[0].adv_images, runs[1].adv_images)
                                                  and runs[0].loss_lists == runs[1].loss_lists))
    open(out_path, "w").write("\n".join(lines) + "\n")


def test_tracked_tuning_file_loads_read_only_and_keeps_the_arithmetic(tmp_path):
    from vqattack_amd.whitebox import tuned_gemms
    if not os.path.exists(tuned_gemms.DEFAULT_FILE):
        pytest.skip("no tuning file tracked")
    bad = tmp_path / "other_stack.csv"
    text = open(tuned_gemms.DEFAULT_FILE).read().replace("Validator,PT_VERSION,", "Validator,PT_VERSION,0.")
    bad.write_text(text)
    out = str(tmp_path / "child.out")
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=_child, args=(out, str(bad)))
    p.start()
    p.join(timeout=600)
    assert not p.is_alive() and p.exitcode == 0
    got = dict(ln.split(" ", 1) for ln in open(out).read().splitlines())
    assert got["bad_file_accepted"] == "False", "a file recorded for another software stack must be ignored"
    assert got["enabled"].startswith("True tuning False") and int(got["enabled"].split()[-1]) > 0
    # another solution of the same library: fp32 GEMM with another summation order
    assert float(got["mm_err"]) <= 1e-5 and float(got["addmm_err"]) <= 1e-5
    assert int(got["entries_checked"]) >= 60
    assert got["entries_unstable"] == "none", "recorded solutions that are not bit-stable run to run: " + got["entries_unstable"]
    assert float(got["entries_worst_err"]) <= 2e-5
    assert got["attack_bitwise_equal"] == "True", "an attack with the recorded GEMM solutions is not reproducible"
