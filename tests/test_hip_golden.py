"""End-to-end parity of the shipped operators (drop-in API -> C ABI -> HIP) against the REFERENCE's outputs.

Every case of ``tests/golden/cases.py`` is run on the MI355X through ``vqattack_amd.dropin`` with the same seeded
inputs and toy white box, and compared with ``tests/golden/cleverhans_golden.npz`` (produced by the reference itself
on CPU).  The white-box forward/backward runs in PyTorch-ROCm, so gradients differ from the CPU's in the last bits;
the stated fp32 tolerance is therefore:

  * L-inf perturbations: every pixel moves by exactly +-eps_iter per step, so a last-bit gradient difference can only
    show where |grad| ~ 0 flips sign.  Required: >= 99.5 % of the pixels BIT-IDENTICAL to the reference and no pixel
    further than 2 * eps_iter * steps from it.
  * L2 / L1 perturbations and text gradients: 1e-4 absolute / 1e-3 relative.
  * losses: 2e-4 relative.
"""
import numpy as np
import pytest
import torch

from tests.adapters import ProductImpl
from tests.golden.cases import ALL_CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
def test_product_matches_reference(case, golden):
    res = run_case(ProductImpl(), case, "cuda:0")
    for key, val in res.items():
        if key == "init_eta":
            continue
        want = golden["{}/{}".format(case["name"], key)]
        got = val.detach().cpu().numpy()
        assert got.shape == want.shape, key
        tag = (case["name"], key)
        if case["op"] in ("clip_eta", "optimize_linear", "zero_out_clipped_grads"):
            if case["norm"] == 2:
                assert np.allclose(got, want, rtol=5e-6, atol=1e-12), tag
            else:
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), tag
        elif key == "adv":
            steps = case.get("steps", 1) * (2 if case.get("ls") == 0 else 1)
            step = case.get("eps_iter", case["eps"])
            if case["norm"] == "inf":
                same = np.mean(got.view(np.uint32) == want.view(np.uint32))
                assert same >= 0.995, tag + (same,)
                assert np.abs(got - want).max() <= 2 * step * steps + 1e-6, tag
            else:
                assert np.allclose(got, want, rtol=1e-3, atol=1e-4), tag + (np.abs(got - want).max(),)
        elif key in ("loss", "loss_list"):
            assert np.allclose(got, want, rtol=2e-4, atol=1e-5), tag + (got, want)
        elif key == "text_grad":
            assert np.allclose(got, want, rtol=1e-3, atol=1e-5), tag + (np.abs(got - want).max(),)
        else:
            raise AssertionError("unchecked output {}".format(tag))


def test_reference_error_behaviour_on_device():
    """Exceptions of the reference's operators (SURVEY.md 8b 'error conventions'), raised before any launch."""
    from vqattack_amd import dropin
    pgd = dropin.load("albef").projected_gradient_descent.projected_gradient_descent
    fgm = dropin.load("albef").fast_gradient_method.fast_gradient_method
    x = torch.zeros(1, 3, 8, 8, device="cuda:0")
    fn = lambda t: [t.reshape(1, 3, 64), t.reshape(1, 3, 64)]          # noqa: E731
    y = [torch.ones(1, 3, 64, device="cuda:0"), torch.ones(1, 3, 64, device="cuda:0")]
    with pytest.raises(NotImplementedError):
        pgd(fn, x, 0.1, 0.01, 1, 1, ori_x=x, ls=1, y=y)
    with pytest.raises(ValueError):
        pgd(fn, x, 0.1, 0.01, 1, 3, ori_x=x, ls=1, y=y)
    with pytest.raises(ValueError):
        pgd(fn, x, -0.1, 0.01, 1, np.inf, ori_x=x, ls=1, y=y)
    with pytest.raises(ValueError):
        pgd(fn, x, 0.1, -0.01, 1, np.inf, ori_x=x, ls=1, y=y)
    with pytest.raises(ValueError):
        pgd(fn, x, 0.1, 0.01, 1, np.inf, clip_min=1, clip_max=-1, ori_x=x, ls=1, y=y)
    with pytest.raises(AssertionError):
        pgd(fn, x, 0.1, 0.2, 1, np.inf, ori_x=x, ls=1, y=y)
    assert pgd(fn, x, 0, 0.01, 1, np.inf, ori_x=x, ls=1, y=y) is x       # bare tensor, not a tuple
    assert pgd(fn, x, 0.1, 0, 1, np.inf, ori_x=x, ls=1, y=y) is x
    assert fgm(fn, x, 0, np.inf, x, y=y, ls=1) is x
    with pytest.raises(ValueError):                                    # one-sided clip is rejected by the FGM step
        fgm(fn, x + 0.5, 0.1, np.inf, x, clip_min=-1, y=list(y), ls=1)
    with pytest.raises(AssertionError):                                # input outside [clip_min, clip_max]
        pgd(fn, x + 2.0, 0.1, 0.01, 1, np.inf, clip_min=-1, clip_max=1, ori_x=x, ls=1, y=list(y))
    adv, losses = pgd(fn, x + 2.0, 0.1, 0.01, 1, np.inf, clip_min=-1, clip_max=1, ori_x=x, ls=1, y=list(y),
                      sanity_checks=False)
    assert len(losses) == 1 and float(adv.max()) <= 1.0
    with pytest.raises(ValueError):
        dropin.load("albef").projected_gradient_descent_vl.projected_gradient_descent(
            fn, [x, x], 0.1, 0.01, 1, np.inf, ori_x=x, ls=0, y=y, attack_mask=[0])
