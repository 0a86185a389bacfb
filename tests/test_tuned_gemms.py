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
    p.join(timeout=300)
    assert not p.is_alive() and p.exitcode == 0
    got = dict(ln.split(" ", 1) for ln in open(out).read().splitlines())
    assert got["bad_file_accepted"] == "False", "a file recorded for another software stack must be ignored"
    assert got["enabled"].startswith("True tuning False") and int(got["enabled"].split()[-1]) > 0
    # another solution of the same library: fp32 GEMM with another summation order
    assert float(got["mm_err"]) <= 1e-5 and float(got["addmm_err"]) <= 1e-5
