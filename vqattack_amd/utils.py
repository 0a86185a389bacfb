"""``cleverhans.torch.utils`` of the reference, on the HIP kernels.

Reference: ``ALBEF_VQAttack/cleverhans/cleverhans/torch/utils.py`` -- ``clip_eta`` :8-40, ``optimize_linear`` :70-128
(the VLMO copy is arithmetically identical).  Same signatures, return values and exceptions, including the
``AssertionError`` of the self-checks inside ``optimize_linear`` (L1: the result has unit L1 norm, L2: unit L2 norm;
:101-104, :110-116): the kernels report the only inputs that can trip them -- an all-zero or NaN L1 gradient, a
non-finite L2 norm -- as a flag bit (``VQA_FLAG_DEGENERATE``), read here once per call like the reference's own ``assert``
does; inside the attack operators the same bit goes to the attack's flag word and is read once per PGD call.
"""
import numpy as np
import torch

from . import ops
from ._hip import VQA_FLAG_DEGENERATE, dev_f32


def _device_tensor(t, name):
    dev_f32(t, name, contiguous=False)
    return t if t.is_contiguous() else t.contiguous()


def clip_eta(eta, norm, eps):
    """Project ``eta`` on the eps-ball.  L-inf: returns a new tensor; L2: rescales ``eta`` IN PLACE and returns it
    (as the reference's ``eta *= factor``); L1: ``NotImplementedError``; other norms: ``ValueError``."""
    if norm not in (np.inf, 1, 2):
        raise ValueError("norm must be np.inf, 1, or 2.")
    if norm == 1:
        raise NotImplementedError("L1 clip is not implemented.")
    if norm == np.inf:
        return ops.clip_eta_linf(_device_tensor(eta, "eta"), eps)
    dev_f32(eta, "eta", contiguous=False)
    work = eta if eta.is_contiguous() else eta.contiguous()
    ops.scale_per_sample(work, ops.sumsq_per_sample(work), None, eps, kind=0, out=work)
    if work is not eta:
        eta.copy_(work)
    return eta


def optimize_linear(grad, eps, norm=np.inf):
    """argmax_{|eta|_norm <= eps} <eta, grad>: eps*sign (inf), eps*grad/|grad|_2 (2), eps*sign*[|g|==max]/ties (1)."""
    if norm == np.inf:
        return ops.optimize_linear_linf(_device_tensor(grad, "grad"), eps)
    if norm in (1, 2):
        g = _device_tensor(grad, "grad")
        flag = ops.new_flag(g.device)
        if norm == 1:
            amax, ties = ops.absmax_ties_per_sample(g)
            out = ops.scale_per_sample(g, amax, ties, eps, kind=2, flag=flag)
        else:
            out = ops.scale_per_sample(g, ops.sumsq_per_sample(g), None, eps, kind=1, flag=flag)
        # the reference's `assert torch.all(opt_pert_norm == 1)` / `assert torch.allclose(...)`: one host read
        assert not int(flag.item()) & VQA_FLAG_DEGENERATE, \
            "optimize_linear: the optimal perturbation does not have unit norm (all-zero or non-finite gradient)"
        return out
    raise NotImplementedError("Only L-inf, L1 and L2 norms are currently implemented.")


def zero_out_clipped_grads(grad, x, clip_min, clip_max):
    """Erase gradient entries whose update would be clipped away (x at a bound and the gradient pointing outwards).
    Reference: utils.py:131-149 (defined there, not called by the attack drivers)."""
    return ops.zero_out_clipped_grads(_device_tensor(grad, "grad"), _device_tensor(x, "x"), clip_min, clip_max)


def get_or_guess_labels(model, x, **kwargs):
    """Labels for crafting an adversarial example: ``y`` (untargeted), ``y_target`` (targeted) or the model's own
    prediction.  Reference: utils.py:43-67 (host logic; unused by the attack drivers)."""
    if "y" in kwargs and "y_target" in kwargs:
        raise ValueError("Can not set both 'y' and 'y_target'.")
    if "y" in kwargs:
        return kwargs["y"]
    if kwargs.get("y_target") is not None:
        return kwargs["y_target"]
    _, labels = torch.max(model(x), 1)
    return labels
