"""Generate ``tests/golden/cleverhans_golden.npz`` from the REFERENCE's own functions.

Runs only in the build container (needs ``/root/reference``); the GPU box never sees the reference.
Nothing of the reference (source, bytecode, pickles) is written to the repo -- only input/output
arrays of the calls listed in ``tests/golden/cases.py``.

    PYTHONDONTWRITEBYTECODE=1 python -m tests.golden.make_golden

Import notes (recorded in DESIGN.md): ``cleverhans.torch.utils`` imports as-is.  The four attack
modules have a top-level ``from torchvision import transforms`` that is only used by the dead
``input_diversity`` helper; torchvision is not installed in this image, so an EMPTY placeholder
module is registered under that name before import.  No reference code is replaced or emulated.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOTS = {"albef": REF + "/ALBEF_VQAttack/cleverhans", "vlmo": REF + "/VLMO_VQAttack/cleverhans"}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cleverhans_golden.npz")


def _load_reference(flavor):
    for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
        del sys.modules[name]
    if "torchvision" not in sys.modules:
        tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    sys.path[:] = [p for p in sys.path if p not in ROOTS.values()]
    sys.path.insert(0, ROOTS[flavor])
    mods = {}
    for short, path in [("utils", "cleverhans.torch.utils"),
                        ("pgd", "cleverhans.torch.attacks.projected_gradient_descent"),
                        ("fgm", "cleverhans.torch.attacks.fast_gradient_method"),
                        ("pgd_vl", "cleverhans.torch.attacks.projected_gradient_descent_vl"),
                        ("fgm_vl", "cleverhans.torch.attacks.fast_gradient_method_vl")]:
        mods[short] = importlib.import_module(path)
        assert mods[short].__file__.startswith(ROOTS[flavor]), mods[short].__file__
    return mods


class ReferenceImpl:
    accepts_init_eta = False

    def __init__(self):
        self._mods = {}
        self._cur = None

    def _m(self, flavor):
        # both copies are top-level packages called `cleverhans`: keep only one loaded at a time
        if self._cur != flavor:
            self._mods = _load_reference(flavor)
            self._cur = flavor
        return self._mods

    def clip_eta(self, eta, norm, eps):
        return self._m("albef")["utils"].clip_eta(eta, norm, eps)

    def optimize_linear(self, grad, eps, norm):
        return self._m("albef")["utils"].optimize_linear(grad, eps, norm)

    def zero_out_clipped_grads(self, grad, x, clip_min, clip_max):
        return self._m("albef")["utils"].zero_out_clipped_grads(grad, x, clip_min, clip_max)

    def fgm(self, flavor):
        return self._m(flavor)["fgm"].fast_gradient_method

    def pgd(self, flavor):
        return self._m(flavor)["pgd"].projected_gradient_descent

    def fgm_vl(self, flavor):
        return self._m(flavor)["fgm_vl"].fast_gradient_method

    def pgd_vl(self, flavor):
        return self._m(flavor)["pgd_vl"].projected_gradient_descent


def main(out=None):
    out = out or OUT
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from tests.golden.cases import ALL_CASES, UTIL_CASES, run_case, util_input

    torch.set_num_threads(1)   # single-threaded reductions: results independent of the host's core count
    impl = ReferenceImpl()
    blob = {}
    for case in ALL_CASES:
        res = run_case(impl, case, "cpu")
        for key, val in res.items():
            blob["{}/{}".format(case["name"], key)] = val.detach().cpu().numpy()
        print("{:34s} {}".format(case["name"], {k: tuple(v.shape) for k, v in res.items()}))
    # the VLMO copy of utils.py must agree with the ALBEF copy on the same inputs
    vl_utils = impl._m("vlmo")["utils"]
    for case in UTIL_CASES:
        if case["op"] == "zero_out_clipped_grads":
            continue
        t = util_input(case)
        norm = np.inf if case["norm"] == "inf" else case["norm"]
        fn = vl_utils.clip_eta if case["op"] == "clip_eta" else None
        got = fn(t.clone(), norm, case["eps"]) if fn else vl_utils.optimize_linear(t.clone(), case["eps"], norm)
        assert np.array_equal(got.numpy(), blob[case["name"] + "/out"], equal_nan=True), case["name"]
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
