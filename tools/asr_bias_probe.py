"""Is the product's adversarial example systematically weaker (or stronger) than the oracle's?

Every success bit that differed so far went the same way (oracle success, product failure), each on a victim tied to a
few 1e-4.  A tie flips with the sign of the difference between the two images' effect on the victim; if that difference
had a nonzero MEAN, the flips would be one-directional.  This probe measures the mean directly on K seeded VLMO-base
samples, full 40-step image attack: product = one batched attack on the GPU, oracle = the per-sample CPU loop
(oracle/cleverhans_cpu + oracle/adapters_ref); for each sample
  * the white-box objective the attack maximises, evaluated on the final adversarial image by ONE scorer (the GPU
    model) for both images:   loss(product image) - loss(oracle image),
  * the victim's signed clean-answer margin on the adversarial pair (logit of the clean answer - best other logit among
    the first `k` classes), GPU scorer for both images:   margin(product image) - margin(oracle image)
    (positive = the product's image leaves the victim closer to / further inside its clean answer).
Prints per-sample lines and the means with their standard errors.  Test infrastructure (imports oracle/).

    python tools/asr_bias_probe.py --n 24 > profiles/r05/asr_bias_probe.jsonl
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--k", type=int, default=4, help="size of the victim's closed answer set")
    ap.add_argument("--seed", type=int, default=151)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import VlmoRefAdapters
    from tests.conftest import usable_cores
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    torch.set_num_threads(usable_cores())
    dev = torch.device("cuda", 0)
    white = FrozenVlmo(vlmo_base(384), seed=3)
    black = FrozenVlmo.finetuned_from(white, seed=4)
    white_gpu, black_gpu = copy.deepcopy(white).to(dev), copy.deepcopy(black).to(dev)
    g = torch.Generator().manual_seed(args.seed)
    n = args.n
    ids = torch.zeros(n, 40, dtype=torch.long)
    for s in range(n):
        w = int(torch.randint(4, 13, (1,), generator=g))
        ids[s, 0], ids[s, 1 + w] = 101, 102
        ids[s, 1:1 + w] = torch.randint(1000, 30522, (w,), generator=g)
    masks = (ids != 0).long()
    img = torch.empty(n, 3, 384, 384).uniform_(-1, 1, generator=g)
    eta = torch.empty_like(img).uniform_(-0.125, 0.125, generator=g)
    adapters = VlmoAttackAdapters(white_gpu)
    attack = BatchedVQAttack(adapters, "vlmo", white_gpu.embedding_tables(), AttackConfig(budget=args.steps))
    res = attack.attack_batch(img.to(dev), ids.to(dev), masks.to(dev), torch.zeros_like(ids, dtype=torch.bool).to(dev),
                              init_eta=eta.to(dev))
    prod = res.adv_images

    def objective(image, s):
        """The white-box feature loss of sample s (what the PGD maximises: fast_gradient_method.py:106-114, oracle
        restatement) on `image`, evaluated by ONE model (the GPU copy, reference-style packing, fp64 sums) for both images."""
        from vqattack_amd.whitebox.reference_style import VlmoReferenceClosures
        me = VlmoReferenceClosures(white_gpu, dict(text_ids=ids[s:s + 1].to(dev), text_masks=masks[s:s + 1].to(dev)))
        y = [t.double() for t in me.Gen_ori_feats(img[s:s + 1].to(dev))]
        with torch.no_grad():
            out = [t.double() for t in me.pgd_attack(image)]
        return float(oracle._vlmo_feature_loss(out, y))

    def margin(image, s, clean):
        with torch.no_grad():
            _, states = black_gpu.encode(image, black_gpu.text_embeddings(ids[s:s + 1].to(dev)), masks[s:s + 1].to(dev))
            lg = black_gpu.vqa_classifier(black_gpu.pooled(states))[0, :args.k].double()
        other = torch.cat([lg[:clean], lg[clean + 1:]]).max()
        return float(lg[clean] - other)

    with torch.no_grad():
        clean = black_gpu.vqa_answer(img.to(dev), ids.to(dev), masks.to(dev), n_answers=args.k).cpu().tolist()
    d_loss, d_margin, same = [], [], []
    for s in range(n):
        ad = VlmoRefAdapters(white, ids[s:s + 1], masks[s:s + 1])
        with torch.enable_grad():
            adv, _ = oracle.projected_gradient_descent(ad.pgd_attack, img[s:s + 1], 0.125, 0.01, args.steps, np.inf,
                                                       clip_min=-1, clip_max=1, y=ad.gen_ori_feats(img[s:s + 1]),
                                                       ori_x=img[s:s + 1], time=0, ls=1, flavor="vlmo",
                                                       init_eta=eta[s:s + 1])
        ora = adv.detach().to(dev)
        lp, lo = objective(prod[s:s + 1], s), objective(ora, s)
        mp, mo = margin(prod[s:s + 1], s, clean[s]), margin(ora, s, clean[s])
        d_loss.append((lp - lo) / abs(lo))
        d_margin.append(mp - mo)
        same.append(float((prod[s:s + 1] == ora).float().mean()))
        print(json.dumps(dict(sample=s, pixels_identical=round(same[-1], 5), loss_product=lp, loss_oracle=lo,
                              rel_loss_diff=d_loss[-1], margin_product=mp, margin_oracle=mo, margin_diff=d_margin[-1])),
              flush=True)

    def stat(x):
        x = np.asarray(x)
        return dict(mean=float(x.mean()), stderr=float(x.std(ddof=1) / len(x) ** 0.5), positive=int((x > 0).sum()),
                    negative=int((x < 0).sum()))
    print(json.dumps(dict(summary=True, n=n, steps=args.steps, k=args.k, pixels_identical_mean=float(np.mean(same)),
                          rel_loss_diff_product_minus_oracle=stat(d_loss),
                          clean_margin_diff_product_minus_oracle=stat(d_margin))), flush=True)


if __name__ == "__main__":
    main()
