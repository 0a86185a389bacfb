"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the VQAttack PGD hot path.

This module is a plain eager PyTorch-CPU restatement of the reference's
cleverhans-style operator API.  It is the *checker* for the HIP path and the
"port" CPU baseline of bench.py; nothing under ``vqattack_amd/`` may import it
(only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do).

Parity pin: every function here is checked bit-for-bit (``torch.equal``)
against outputs of the reference's own functions, generated in the build
container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (see ``tests/test_oracle_golden.py``).  The reference's
own known-answer tests for ``clip_eta`` / ``optimize_linear``
(``VLMO_VQAttack/cleverhans/cleverhans/torch/tests/test_utils.py:23-111``) are
restated in ``tests/test_oracle_kat.py``.

Reference files followed (relative to the reference root):
  A-ch = ALBEF_VQAttack/cleverhans/cleverhans/torch
  V-ch = VLMO_VQAttack/cleverhans/cleverhans/torch
  * clip_eta ................. A-ch/utils.py:8-40      (= V-ch/utils.py:8-40)
  * optimize_linear .......... A-ch/utils.py:70-128
  * fast_gradient_method ..... A-ch/attacks/fast_gradient_method.py:30-165
                               V-ch/attacks/fast_gradient_method.py:36-152
  * projected_gradient_descent A-ch/attacks/projected_gradient_descent.py:10-199
                               V-ch/attacks/projected_gradient_descent.py:10-196
  * fast_gradient_method (vl)  A-ch/attacks/fast_gradient_method_vl.py:30-130
                               V-ch/attacks/fast_gradient_method_vl.py:34-141
  * projected_gradient_descent (vl)
                               A-ch/attacks/projected_gradient_descent_vl.py:10-168

The op order of the reference's unfused eager chain is kept on purpose: the
point of the oracle is to reproduce the reference's floating-point results, not
to be fast.  ``flavor`` selects the ALBEF or the VLMO copy where they differ.
"""
import numpy as np
import torch
import torch.nn.functional as F

MLM_VOCAB = 30522  # hard-coded in the reference (A-ch/attacks/fast_gradient_method.py:103)
ALBEF = "albef"
VLMO = "vlmo"


# --------------------------------------------------------------------------- utils
def clip_eta(eta, norm, eps):
    """A-ch/utils.py:8-40.  L-inf: clamp; L2: per-sample rescale IN PLACE; L1: raises."""
    if norm not in (np.inf, 1, 2):
        raise ValueError("norm must be np.inf, 1, or 2.")
    if norm == np.inf:
        return torch.clamp(eta, -eps, eps)
    if norm == 1:
        raise NotImplementedError("L1 clip is not implemented.")
    tiny = torch.tensor(1e-12, dtype=eta.dtype, device=eta.device)
    dims = list(range(1, eta.dim()))
    length = torch.sqrt(torch.max(tiny, torch.sum(eta ** 2, dim=dims, keepdim=True)))
    factor = torch.min(torch.tensor(1.0, dtype=eta.dtype, device=eta.device), eps / length)
    eta *= factor
    return eta


def optimize_linear(grad, eps, norm=np.inf):
    """A-ch/utils.py:70-128 (including its two self-check asserts)."""
    dims = list(range(1, grad.dim()))
    tiny = torch.tensor(1e-12, dtype=grad.dtype, device=grad.device)
    if norm == np.inf:
        direction = torch.sign(grad)
    elif norm == 1:
        mag = torch.abs(grad)
        sgn = torch.sign(grad)
        keep = [grad.size(0)] + [1] * (grad.dim() - 1)
        top, _ = torch.max(mag.view(grad.size(0), -1), 1)
        at_top = mag.eq(top.view(keep)).to(torch.float)
        ties = at_top
        for d in dims:
            ties = torch.sum(ties, d, keepdim=True)
        direction = sgn * at_top / ties
        l1 = direction.abs().sum(dim=dims)
        assert torch.all(l1 == torch.ones_like(l1))
    elif norm == 2:
        sq = torch.max(tiny, torch.sum(grad ** 2, dims, keepdim=True))
        direction = grad / torch.sqrt(sq)
        l2 = direction.pow(2).sum(dim=dims, keepdim=True).sqrt()
        expect = (sq <= tiny).to(torch.float) * l2 + (sq > tiny).to(torch.float)
        assert torch.allclose(l2, expect, rtol=1e-05, atol=1e-08)
    else:
        raise NotImplementedError("Only L-inf, L1 and L2 norms are currently implemented.")
    return eps * direction


def zero_out_clipped_grads(grad, x, clip_min, clip_max):
    """A-ch/utils.py:131-149."""
    sg = torch.sign(grad)
    low = torch.le(x, clip_min) & torch.lt(sg, 0)
    high = torch.ge(x, clip_max) & torch.gt(sg, 0)
    return torch.where(low | high, torch.zeros_like(grad), grad)


def get_or_guess_labels(model, x, **kwargs):
    """A-ch/utils.py:43-67."""
    if "y" in kwargs and "y_target" in kwargs:
        raise ValueError("Can not set both 'y' and 'y_target'.")
    if "y" in kwargs:
        return kwargs["y"]
    if "y_target" in kwargs and kwargs["y_target"] is not None:
        return kwargs["y_target"]
    _, labels = torch.max(model(x), 1)
    return labels


# --------------------------------------------------------------------------- losses
def _mlm_ce(logits, labels):
    """CE over the MLM vocabulary, 2-d labels or a sum over K label sets (3-d labels).
    A-ch/attacks/fast_gradient_method.py:131-142, V-ch/...:116-126."""
    flat = logits.view(-1, MLM_VOCAB)
    if labels.dim() == 2:
        return F.cross_entropy(flat, labels.view(-1), ignore_index=-100)
    if labels.dim() == 3:
        loss = F.cross_entropy(flat, labels[:, 0, :].view(-1), ignore_index=-100)
        for k in range(labels.size(1) - 1):
            loss += F.cross_entropy(flat, labels[:, k + 1, :].view(-1), ignore_index=-100)
        return loss
    raise ValueError


def _albef_feature_loss(out, y):
    """A-ch/attacks/fast_gradient_method.py:120-127.  Truncates out/y IN PLACE (lists)."""
    cos_tok = torch.nn.CosineSimilarity(dim=2, eps=1e-6)
    n1 = min(out[1].shape[1], y[1].shape[1])
    out[1] = out[1][:, :n1, :]
    y[1] = y[1][:, :n1, :]
    n0 = min(out[0].shape[1], y[0].shape[1])
    out[0] = out[0][:, :n0, :]
    y[0] = y[0][:, :n0, :]
    return torch.sum(torch.sum(-cos_tok(out[1], y[1]), 1) + torch.sum(-cos_tok(out[0], y[0]), 1))


def _vlmo_feature_loss(out, y):
    """V-ch/attacks/fast_gradient_method.py:106-114."""
    cos_row = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
    cos_tok = torch.nn.CosineSimilarity(dim=2, eps=1e-6)
    if out[2].shape[1] != y[2].shape[1]:
        n = min(out[2].shape[1], y[2].shape[1])
        out[2] = out[2][:, :n, :]
        y[2] = y[2][:, :n, :]
    return torch.sum(-cos_row(out[1], y[1]) + torch.sum(-cos_tok(out[2], y[2]), 1))


def _check_fgm_args(norm, eps, clip_min, clip_max):
    if norm not in (np.inf, 1, 2):
        raise ValueError("Norm order must be either np.inf, 1, or 2, got {} instead.".format(norm))
    if eps < 0:
        raise ValueError("eps must be greater than or equal to 0, got {} instead".format(eps))
    if clip_min is not None and clip_max is not None and clip_min > clip_max:
        raise ValueError("clip_min must be less than or equal to clip_max")


def _range_flags(x, clip_min, clip_max):
    flags = []
    if clip_min is not None:
        flags.append(torch.all(torch.ge(x, torch.tensor(clip_min, device=x.device, dtype=x.dtype))))
    if clip_max is not None:
        flags.append(torch.all(torch.le(x, torch.tensor(clip_max, device=x.device, dtype=x.dtype))))
    return flags


def _finish_step(x, grad, eps, norm, clip_min, clip_max):
    adv = x + optimize_linear(grad, eps, norm)
    if clip_min is not None or clip_max is not None:
        if clip_min is None or clip_max is None:
            raise ValueError("one-sided clipping is not supported")
        adv = torch.clamp(adv, clip_min, clip_max)
    return adv


# --------------------------------------------------------------------------- FGM
def fast_gradient_method(model_fn, x, eps, norm, ori_x, clip_min=None, clip_max=None, y=None,
                         targeted=False, sanity_checks=False, ls=None, bkp=None, bkp_y=None,
                         flavor=ALBEF):
    """One FGM step.  Returns ``(adv_x, loss)``; bare ``x`` when ``eps == 0``."""
    _check_fgm_args(norm, eps, clip_min, clip_max)
    if eps == 0:
        return x
    flags = _range_flags(x, clip_min, clip_max)
    x = x.clone().detach().to(torch.float).requires_grad_(True)
    out = model_fn(x)

    if flavor == ALBEF:
        if ls == 0:  # MLM label/logit length mismatch -> fall back to the feature loss
            rows = out[0].view(-1, MLM_VOCAB).shape[0]
            lab = y[0]
            bad = False
            if lab.dim() == 2:
                bad = lab.view(-1).shape[0] != rows
            elif lab.dim() == 3:
                bad = any(lab[:, k, :].contiguous().view(-1).shape[0] != rows
                          for k in range(lab.shape[1]))
            if bad:
                ls, out, y = 1, bkp(x), bkp_y
        if ls == 1:
            loss = _albef_feature_loss(out, y)
        elif ls == 0:
            loss = _mlm_ce(out[0], y[0])
    else:
        if ls == 1:
            loss = _vlmo_feature_loss(out, y)
        elif ls == 0:
            loss = _mlm_ce(out[0], y[0])
        else:  # mixed loss, V-ch/attacks/fast_gradient_method.py:127-131
            cos_row = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
            cos_tok = torch.nn.CosineSimilarity(dim=2, eps=1e-6)
            loss = torch.sum(-cos_row(out[1], y[1]) + torch.sum(-cos_tok(out[2], y[2]), 1)) \
                / (out[2].shape[0] * out[2].shape[1]) \
                + 0.1 * F.cross_entropy(out[0].view(-1, MLM_VOCAB), y[0].view(-1), ignore_index=-100)
            for syn in y[3]:
                loss = loss + 0.1 * F.cross_entropy(out[0].view(-1, MLM_VOCAB), syn[0].view(-1),
                                                    ignore_index=-100)
    if targeted:
        loss = -loss
    loss.requires_grad_(True).backward()
    adv = _finish_step(x, x.grad, eps, norm, clip_min, clip_max)
    if sanity_checks:
        assert np.all(flags)
    return adv, loss


def fast_gradient_method_vl(model_fn, x, eps, norm, ori_x, clip_min=None, clip_max=None, y=None,
                            targeted=False, sanity_checks=False, ls=None, text_emb_pick=None,
                            flavor=ALBEF):
    """FGM on ``x = [image, text_embeds]``; returns ``(adv_image, text_grad[:, pick])``."""
    _check_fgm_args(norm, eps, clip_min, clip_max)
    if eps == 0:
        return x
    flags = _range_flags(x[0], clip_min, clip_max)
    x[0] = x[0].clone().detach().to(torch.float).requires_grad_(True)
    x[1] = x[1].clone().detach().to(torch.float).requires_grad_(True)
    out = model_fn([x[0], x[1]])
    if flavor == ALBEF:
        cos_tok = torch.nn.CosineSimilarity(dim=2, eps=1e-6)
        n1 = min(out[1].shape[1], y[1].shape[1])
        out[1] = out[1][:, :n1, :]
        y[1] = y[1][:, :n1, :]
        n0 = min(out[0].shape[1], y[0].shape[1])
        out[0] = out[0][:, :n0, :]
        y[0] = y[0][:, :n0, :]
        if ls == 1:
            loss = torch.sum(torch.sum(-cos_tok(out[0], y[0]), 1) + torch.sum(-cos_tok(out[1], y[1]), 1))
        elif ls == 0:
            loss = F.cross_entropy(out[0].view(-1, MLM_VOCAB), y[0].view(-1), ignore_index=-100)
    else:
        cos_row = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
        cos_tok = torch.nn.CosineSimilarity(dim=2, eps=1e-6)
        n = min(out[2].shape[1], y[2].shape[1])
        out[2] = out[2][:, :n, :]
        y[2] = y[2][:, :n, :]
        if ls == 1:
            loss = torch.sum(-cos_row(out[1], y[1]) + torch.sum(-cos_tok(out[2], y[2]), 1))
        elif ls == 0:
            loss = F.cross_entropy(out[0].view(-1, MLM_VOCAB), y[0].view(-1), ignore_index=-100)
    if targeted:
        loss = -loss
    loss.requires_grad_(True).backward()
    direction = optimize_linear(x[0].grad, eps, norm)
    adv = x[0] + direction
    text_grad = x[1].grad[:, text_emb_pick]
    if clip_min is not None or clip_max is not None:
        if clip_min is None or clip_max is None:
            raise ValueError("one-sided clipping is not supported")
        adv = torch.clamp(adv, clip_min, clip_max)
    if sanity_checks:
        assert np.all(flags)
    return adv, text_grad


# --------------------------------------------------------------------------- PGD
def _check_pgd_args(norm, eps, eps_iter, clip_min, clip_max):
    """Returns True when the caller must hand back the bare input (eps or eps_iter == 0)."""
    if norm == 1:
        raise NotImplementedError("PGD with norm=1 is not enabled in the reference.")
    if norm not in (np.inf, 2):
        raise ValueError("Norm order must be either np.inf or 2.")
    if eps < 0:
        raise ValueError("eps must be greater than or equal to 0, got {} instead".format(eps))
    if eps == 0:
        return True
    if eps_iter < 0:
        raise ValueError("eps_iter must be greater than or equal to 0, got {} instead".format(eps_iter))
    if eps_iter == 0:
        return True
    assert eps_iter <= eps, (eps_iter, eps)
    if clip_min is not None and clip_max is not None and clip_min > clip_max:
        raise ValueError("clip_min must be less than or equal to clip_max")
    return False


def _start_point(x, norm, eps, clip_min, clip_max, time, rand_minmax, init_eta):
    if time == 0:  # the reference derives rand_init from `time`, ignoring the kwarg
        if init_eta is not None:
            eta = init_eta.clone()
        else:
            eta = torch.zeros_like(x).uniform_(-(rand_minmax if rand_minmax is not None else eps),
                                               (rand_minmax if rand_minmax is not None else eps))
    else:
        eta = torch.zeros_like(x)
    eta = clip_eta(eta, norm, eps)
    adv = x + eta
    if clip_min is not None or clip_max is not None:
        adv = torch.clamp(adv, clip_min, clip_max)
    return adv


def _project(adv, ori_x, norm, eps, clip_min, clip_max):
    eta = clip_eta(adv - ori_x, norm, eps)
    adv = ori_x + eta
    if clip_min is not None or clip_max is not None:
        adv = torch.clamp(adv, clip_min, clip_max)
    return adv


def _final_asserts(flags, eps, eps_iter, norm, clip_min, clip_max, sanity_checks):
    flags = list(flags)
    flags.append(eps_iter <= eps)
    if norm == np.inf and clip_min is not None:
        flags.append(eps + clip_min <= clip_max)
    flags = [f.cpu() for f in flags if f is not True]
    if sanity_checks:
        assert np.all(flags)


def projected_gradient_descent(model_fn, x, eps, eps_iter, nb_iter, norm, clip_min=None, clip_max=None,
                               y=None, ori_x=None, time=None, targeted=False, rand_init=True,
                               rand_minmax=None, sanity_checks=True, ls=None, flavor=ALBEF,
                               init_eta=None):
    """PGD loop.  Returns ``(adv_x, loss_list)``; bare ``x`` when eps/eps_iter == 0.

    ``init_eta`` (oracle extension): the uniform draw to use when ``time == 0`` so that a
    caller can inject the exact perturbation the reference drew from torch's global RNG.
    """
    if _check_pgd_args(norm, eps, eps_iter, clip_min, clip_max):
        return x
    flags = _range_flags(x, clip_min, clip_max)
    adv = _start_point(x, norm, eps, clip_min, clip_max, time, rand_minmax, init_eta)
    if y is None:
        _, y = torch.max(model_fn(x), 1)
    losses = []
    ori_x = ori_x.requires_grad_(False)
    common = dict(clip_min=clip_min, clip_max=clip_max, targeted=targeted, flavor=flavor)
    for _ in range(nb_iter):
        if ls == 1:
            adv, loss = fast_gradient_method(model_fn, adv, eps_iter, norm, ori_x, y=y, ls=ls, **common)
            losses.append(float(loss.detach().cpu().numpy()))
        else:  # dual loss: feature step, then MLM step on the un-projected result
            if flavor == ALBEF:
                y_feat, y_mlm = [y[1], y[2]], [y[0]]
                extra = dict(bkp=model_fn[0], bkp_y=[y[1], y[2]])
            else:
                y_feat, y_mlm, extra = y, y, {}
            adv, loss = fast_gradient_method(model_fn[0], adv, eps_iter, norm, ori_x, y=y_feat, ls=1,
                                             **common)
            losses.append(float(loss.detach().cpu().numpy()))
            adv, loss = fast_gradient_method(model_fn[1], adv, eps_iter, norm, ori_x, y=y_mlm, ls=0,
                                             **common, **extra)
            losses.append(float(loss.detach().cpu().numpy()))
        adv = _project(adv, ori_x, norm, eps, clip_min, clip_max)
    _final_asserts(flags, eps, eps_iter, norm, clip_min, clip_max, sanity_checks)
    return adv, losses


def projected_gradient_descent_vl(model_fn, x, eps, eps_iter, nb_iter, norm, clip_min=None,
                                  clip_max=None, y=None, ori_x=None, time=None, targeted=False,
                                  rand_init=True, rand_minmax=None, sanity_checks=True, ls=None,
                                  attack_mask=None, flavor=ALBEF, init_eta=None):
    """PGD on ``x = [image, text_embeds]``; returns ``(adv_image, last text gradient rows)``."""
    if _check_pgd_args(norm, eps, eps_iter, clip_min, clip_max):
        return x
    flags = _range_flags(x[0], clip_min, clip_max)
    adv = _start_point(x[0], norm, eps, clip_min, clip_max, time, rand_minmax, init_eta)
    if y is None:
        _, y = torch.max(model_fn(x), 1)
    ori_x = ori_x.requires_grad_(False)
    text_grad = None
    if ls != 1:
        raise ValueError
    for _ in range(nb_iter):
        adv, text_grad = fast_gradient_method_vl(model_fn, [adv, x[1]], eps_iter, norm, ori_x,
                                                 clip_min=clip_min, clip_max=clip_max, y=y,
                                                 targeted=targeted, ls=ls, text_emb_pick=attack_mask,
                                                 flavor=flavor)
        adv = _project(adv, ori_x, norm, eps, clip_min, clip_max)
    _final_asserts(flags, eps, eps_iter, norm, clip_min, clip_max, sanity_checks)
    return adv, text_grad


# --------------------------------------------------------------------------- kernel-level checkers
def fgm_update_given_grad(x, grad, eps, norm, clip_min=None, clip_max=None):
    """Rows 5-8 of SURVEY.md section 2.3 for a GIVEN gradient: clamp(x + optimize_linear(grad, eps, norm))."""
    return _finish_step(x, grad, eps, norm, clip_min, clip_max)


def pgd_tail_given_grad(x, grad, ori_x, eps_iter, eps, norm, clip_min=None, clip_max=None):
    """Rows 5-8 + 10-13 for a GIVEN gradient: the FGM update followed by the eps-ball projection, i.e. what the
    reference computes between two model calls (fast_gradient_method.py:151-160 + projected_gradient_descent.py:146-151)."""
    return _project(_finish_step(x, grad, eps_iter, norm, clip_min, clip_max), ori_x, norm, eps, clip_min, clip_max)


def start_point(x, eta, norm, eps, clip_min=None, clip_max=None):
    """projected_gradient_descent.py:117-120 for a given eta (None = zeros)."""
    eta = torch.zeros_like(x) if eta is None else eta.clone()
    adv = x + clip_eta(eta, norm, eps)
    if clip_min is not None or clip_max is not None:
        adv = torch.clamp(adv, clip_min, clip_max)
    return adv


def range_ok(x, clip_min, clip_max):
    """The sanity flags of projected_gradient_descent.py:95-105 as one bool."""
    return bool(np.all([bool(f) for f in _range_flags(x, clip_min, clip_max)]))
