"""Case registry shared by the golden generator (reference), the oracle tests and the HIP parity tests.

A case fixes the inputs (all derived from seeded numpy ``RandomState`` draws and the toy white-box
of ``tests/toy_models.py``) and names one operator call.  ``run_case(impl, case, device)`` performs
the call through an *implementation adapter* and returns ``{name: tensor | list[float]}``; the same
function therefore drives the reference (in ``make_golden.py``), the CPU oracle and the HIP product.

An adapter exposes, per flavor ("albef" / "vlmo"):
    clip_eta(eta, norm, eps)                          optimize_linear(grad, eps, norm)
    fgm(flavor)(model_fn, x, eps, norm, ori_x, **kw)   pgd(flavor)(model_fn, x, eps, eps_iter, nb_iter, norm, **kw)
    fgm_vl(flavor)(...)                               pgd_vl(flavor)(...)
and ``accepts_init_eta`` (True when ``pgd`` takes the injected uniform draw as ``init_eta=``;
the reference instead draws it from torch's global CPU RNG, seeded by ``case['seed']``).
"""
import numpy as np
import torch

from tests.toy_models import ToyWhiteBox, toy_inputs

INF = float("inf")


def _norm(v):
    return np.inf if v == "inf" else v


UTIL_CASES = [
    dict(name="clip_eta_linf", op="clip_eta", norm="inf", eps=0.5, seed=101),
    dict(name="clip_eta_l2", op="clip_eta", norm=2, eps=0.5, seed=102),
    dict(name="clip_eta_l2_large_eps", op="clip_eta", norm=2, eps=50.0, seed=103),
    dict(name="optlin_linf", op="optimize_linear", norm="inf", eps=0.01, seed=104),
    dict(name="optlin_l2", op="optimize_linear", norm=2, eps=0.3, seed=105),
    dict(name="optlin_l1", op="optimize_linear", norm=1, eps=1.5, seed=106),
    dict(name="zero_clipped", op="zero_out_clipped_grads", norm="inf", eps=0.0, seed=107),
]

# NB: no case starts a time != 0 run exactly at the clean image: there out == y, the cosine loss sits at its
# maximum and the gradient is pure rounding noise, so its sign -- hence the trajectory -- is not reproducible
# across devices.  The reference avoids that point itself by drawing a random start for the first block (time=0).
ATTACK_CASES = [
    # BASELINE.json configs[0]: 4 images x 10 steps, eps = 8/255 (step 2/255), clip [-1, 1]
    dict(name="albef_pgd_linf_cfg0", flavor="albef", op="pgd", batch=4, steps=10, eps=8 / 255,
         eps_iter=2 / 255, norm="inf", ls=1, time=0, start_inside=False, seed=1234),
    dict(name="albef_pgd_linf_ref_literals", flavor="albef", op="pgd", batch=2, steps=6, eps=0.125,
         eps_iter=0.01, norm="inf", ls=1, time=1, start_inside=True, y_extra=2),
    dict(name="albef_pgd_linf_randinit", flavor="albef", op="pgd", batch=2, steps=4, eps=0.125,
         eps_iter=0.01, norm="inf", ls=1, time=0, start_inside=False, seed=77),
    dict(name="albef_pgd_l2", flavor="albef", op="pgd", batch=3, steps=5, eps=2.0, eps_iter=0.5,
         norm=2, ls=1, time=1, start_inside=True),
    dict(name="albef_pgd_dual", flavor="albef", op="pgd", batch=2, steps=3, eps=0.125, eps_iter=0.01,
         norm="inf", ls=0, time=1, start_inside=True, labels="2d"),
    dict(name="albef_pgd_dual_3d", flavor="albef", op="pgd", batch=1, steps=2, eps=0.125,
         eps_iter=0.01, norm="inf", ls=0, time=1, start_inside=True, labels="3d"),
    dict(name="albef_pgd_dual_fallback", flavor="albef", op="pgd", batch=2, steps=2, eps=0.125,
         eps_iter=0.01, norm="inf", ls=0, time=1, start_inside=True, labels="mismatch"),
    dict(name="albef_fgm_linf", flavor="albef", op="fgm", batch=2, eps=0.01, norm="inf", ls=1),
    dict(name="albef_fgm_linf_targeted", flavor="albef", op="fgm", batch=2, eps=0.01, norm="inf", ls=1,
         targeted=True),
    dict(name="albef_fgm_l2", flavor="albef", op="fgm", batch=2, eps=0.7, norm=2, ls=1),
    dict(name="albef_fgm_l1", flavor="albef", op="fgm", batch=2, eps=0.7, norm=1, ls=1),
    dict(name="albef_fgm_mlm", flavor="albef", op="fgm", batch=2, eps=0.01, norm="inf", ls=0,
         labels="2d"),
    dict(name="albef_fgm_noclip", flavor="albef", op="fgm", batch=2, eps=0.01, norm="inf", ls=1,
         clip=False),
    dict(name="albef_pgdvl", flavor="albef", op="pgd_vl", batch=2, steps=1, eps=0.125, eps_iter=0.01,
         norm="inf", ls=1, time=1, start_inside=True, mask=[1, 3, 4]),
    dict(name="albef_pgdvl_2step", flavor="albef", op="pgd_vl", batch=1, steps=2, eps=0.125,
         eps_iter=0.01, norm="inf", ls=1, time=1, start_inside=True, mask=[2]),
    dict(name="vlmo_pgd_linf", flavor="vlmo", op="pgd", batch=1, steps=8, eps=0.125, eps_iter=0.01,
         norm="inf", ls=1, time=1, start_inside=True, y_extra=1),
    dict(name="vlmo_pgd_dual", flavor="vlmo", op="pgd", batch=1, steps=3, eps=0.125, eps_iter=0.01,
         norm="inf", ls=0, time=1, start_inside=True, labels="2d"),
    dict(name="vlmo_fgm_linf", flavor="vlmo", op="fgm", batch=1, eps=0.01, norm="inf", ls=1),
    dict(name="vlmo_fgm_mlm3d", flavor="vlmo", op="fgm", batch=1, eps=0.01, norm="inf", ls=0,
         labels="3d"),
    dict(name="vlmo_fgm_mixed", flavor="vlmo", op="fgm", batch=1, eps=0.01, norm="inf", ls=2,
         labels="2d"),
    dict(name="vlmo_pgdvl", flavor="vlmo", op="pgd_vl", batch=1, steps=1, eps=0.125, eps_iter=0.01,
         norm="inf", ls=1, time=1, start_inside=True, mask=[1, 2, 5]),
    # fast_gradient_method of the *_vl modules called directly (row a6): image step + text_embeds.grad[:, text_emb_pick]
    dict(name="albef_fgmvl", flavor="albef", op="fgm_vl", batch=2, eps=0.01, norm="inf", ls=1, mask=[0, 2, 5]),
    dict(name="albef_fgmvl_targeted_l2", flavor="albef", op="fgm_vl", batch=2, eps=0.7, norm=2, ls=1, mask=[3],
         targeted=True),
    dict(name="vlmo_fgmvl", flavor="vlmo", op="fgm_vl", batch=1, eps=0.01, norm="inf", ls=1, mask=[1, 4]),
]

ALL_CASES = UTIL_CASES + ATTACK_CASES


def case_by_name(name):
    for c in ALL_CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)


# ----------------------------------------------------------------------------- inputs
def util_input(case, device="cpu"):
    r = np.random.RandomState(case["seed"])
    t = r.standard_normal((6, 3, 5, 4)).astype(np.float32)
    t[1] = 0.0                       # an all-zero sample (avoid_zero_div branch)
    t[2, 0, 0, :3] = 0.0             # exact zeros inside a sample (sign(0) = 0)
    t[2, 0, 1, 0] = -0.0
    t[3] *= 1e-8                     # tiny norm
    t[4, 1, 2, 3] = 7.5              # unique max (L1)
    t[5, 0, 0, 0] = 9.0              # a tie (L1)
    t[5, 2, 4, 3] = -9.0
    if case["norm"] == 1:            # the reference's L1 self-check asserts on an all-zero sample
        t[1] = r.standard_normal(t[1].shape).astype(np.float32)
    return torch.from_numpy(t).to(device)


def _labels(kind, batch, length, device):
    r = np.random.RandomState(5)
    if kind == "2d":
        lab = r.randint(0, 30522, (batch, length))
        lab[:, ::2] = -100
        return torch.from_numpy(lab).long().to(device)
    if kind == "3d":
        lab = r.randint(0, 30522, (batch, 3, length))
        lab[:, :, 1::2] = -100
        return torch.from_numpy(lab).long().to(device)
    if kind == "mismatch":           # one token too many -> the ALBEF copy falls back to the feature loss
        lab = r.randint(0, 30522, (batch, length + 1))
        return torch.from_numpy(lab).long().to(device)
    raise KeyError(kind)


def attack_inputs(case, device="cpu"):
    toy = ToyWhiteBox(device=device)
    x0, eta = toy_inputs(case["batch"], eps=case["eps"] if case["op"] not in ("fgm", "fgm_vl") else 0.05)
    x0, eta = x0.to(device), eta.to(device)
    start = torch.clamp(x0 + eta, -1, 1) if case.get("start_inside", True) else x0.clone()
    extra = case.get("y_extra", 0)
    with torch.no_grad():
        if case["flavor"] == "albef":
            txt, img = toy.albef_feats(x0)
            if extra:                # reference truncates to the common token length
                img = torch.cat([img, img[:, :extra]], dim=1)
                txt = torch.cat([txt, txt[:, :1]], dim=1)
            feats = [txt, img]
        else:
            pooled, cls_layers, fl = toy.vlmo_feats(x0)
            if extra:
                fl = torch.cat([fl, fl[:, :extra]], dim=1)
            feats = [pooled, cls_layers, fl]
    return toy, x0, start, feats


# ----------------------------------------------------------------------------- driver
def run_case(impl, case, device="cpu"):
    norm = _norm(case["norm"])
    if case["op"] == "clip_eta":
        t = util_input(case, device)
        return {"out": impl.clip_eta(t.clone(), norm, case["eps"])}
    if case["op"] == "optimize_linear":
        t = util_input(case, device)
        return {"out": impl.optimize_linear(t.clone(), case["eps"], norm)}
    if case["op"] == "zero_out_clipped_grads":
        g = util_input(case, device)
        r = np.random.RandomState(case["seed"] + 1)
        x = torch.from_numpy(r.choice(np.array([-1.0, -0.5, 0.0, 1.0, 1.5, -2.0], dtype=np.float32),
                                      size=tuple(g.shape))).to(device)
        return {"out": impl.zero_out_clipped_grads(g, x, -1.0, 1.0)}

    toy, x0, start, feats = attack_inputs(case, device)
    flavor = case["flavor"]
    clip = dict(clip_min=-1, clip_max=1) if case.get("clip", True) else {}
    ls = case["ls"]
    lab = _labels(case["labels"], case["batch"], toy.text_len, device) if "labels" in case else None

    if flavor == "albef":
        feat_fn, vl_fn, mlm_fn = toy.albef_feats, toy.albef_feats_vl, toy.mlm_logits
        y_feat = [feats[0], feats[1], None, None, None]
        y_dual = [lab, feats[0], feats[1]]
        y_mlm = [lab]
    else:
        feat_fn, vl_fn, mlm_fn = toy.vlmo_feats, toy.vlmo_feats_vl, toy.mlm_logits
        y_feat = [feats[0], feats[1], feats[2]]
        y_dual = [lab, feats[1], feats[2]]
        y_mlm = [lab]

    if case["op"] == "fgm":
        kw = dict(y=None, ls=ls, targeted=case.get("targeted", False), **clip)
        if ls == 1:
            fn, kw["y"] = feat_fn, y_feat
        elif ls == 0:
            fn, kw["y"] = mlm_fn, y_mlm
        else:                        # VLMO mixed loss: y = [labels, cls, feats, synonym label sets]
            fn = toy.vlmo_mixed
            syn = [[_labels("2d", case["batch"], toy.text_len, device).roll(1, 1)],
                   [_labels("2d", case["batch"], toy.text_len, device).roll(2, 1)]]
            kw["y"] = [lab, feats[1], feats[2], syn]
        adv, loss = impl.fgm(flavor)(fn, start, case["eps"], norm, x0, **kw)
        return {"adv": adv.detach(), "loss": loss.detach().reshape(1)}

    if case["op"] == "fgm_vl":
        emb = toy.embed_text(toy.text_ids.expand(case["batch"], -1)).clone()
        adv, tgrad = impl.fgm_vl(flavor)(vl_fn, [start, emb], case["eps"], norm, x0, y=y_feat, ls=ls,
                                          targeted=case.get("targeted", False), text_emb_pick=case["mask"], **clip)
        return {"adv": adv.detach(), "text_grad": tgrad.detach()}

    kw = dict(ori_x=x0, time=case["time"], ls=ls, **clip)
    init_eta = None
    if case["time"] == 0:
        torch.manual_seed(case["seed"])
        init_eta = torch.zeros(start.shape).uniform_(-case["eps"], case["eps"]).to(device)
        if impl.accepts_init_eta:
            kw["init_eta"] = init_eta
        else:
            torch.manual_seed(case["seed"])   # the reference draws the same numbers itself

    if case["op"] == "pgd":
        if ls == 1:
            fn, kw["y"] = feat_fn, y_feat
        else:
            fn, kw["y"] = [feat_fn, mlm_fn], y_dual
        adv, losses = impl.pgd(flavor)(fn, start, case["eps"], case["eps_iter"], case["steps"], norm, **kw)
        out = {"adv": adv.detach(), "loss_list": torch.tensor(losses, dtype=torch.float64)}
        if init_eta is not None:
            out["init_eta"] = init_eta
        return out

    if case["op"] == "pgd_vl":
        emb = toy.embed_text(toy.text_ids.expand(case["batch"], -1)).clone()
        adv, tgrad = impl.pgd_vl(flavor)(vl_fn, [start, emb], case["eps"], case["eps_iter"],
                                          case["steps"], norm, y=y_feat, attack_mask=case["mask"], **kw)
        return {"adv": adv.detach(), "text_grad": tgrad.detach()}
    raise KeyError(case["op"])
