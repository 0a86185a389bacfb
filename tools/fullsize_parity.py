"""Perturbation parity at the metric's full model size: the shipped HIP path on the MI355X vs the CPU oracle.

VLMO-base (12 x 768, 384 px, 40-token questions), 2 images, 8 PGD steps from a shared random start: the product attacks
them as one batch (LayerFeatures, row weights, trimmed padding, fused kernels); the oracle attacks them one by one with
reference-style packing on torch CPU.  Reports per sample: fraction of bit-identical pixels, largest deviation, and
the loss trajectories.  (fp32 GEMMs/attention reduce in a different order on the two devices, so gradients differ in
the last bits and a pixel whose gradient is ~0 can step the other way.)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cleverhans_cpu as oracle  # noqa: E402
from oracle.adapters_ref import VlmoRefAdapters  # noqa: E402
from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack  # noqa: E402
from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base  # noqa: E402

STEPS = int(os.environ.get("STEPS", "8"))
cfg = vlmo_base(384)
dev = torch.device("cuda", 0)
gpu_model, cpu_model = FrozenVlmo(cfg, seed=0).to(dev), FrozenVlmo(cfg, seed=0)
ids = torch.zeros(2, 40, dtype=torch.long)
ids[0, :7] = torch.tensor([101, 2054, 3609, 2003, 1996, 4937, 102])
ids[1, :9] = torch.tensor([101, 2129, 2116, 6077, 2024, 1999, 1996, 2833, 102])
masks = (ids != 0).long()
g = torch.Generator().manual_seed(0)
img = torch.empty(2, 3, 384, 384).uniform_(-1, 1, generator=g)
eta = torch.empty_like(img).uniform_(-0.125, 0.125, generator=g)
attack = BatchedVQAttack(VlmoAttackAdapters(gpu_model), "vlmo", gpu_model.embedding_tables(),
                         AttackConfig(budget=STEPS, sanity_checks=True))
res = attack.attack_batch(img.to(dev), ids.to(dev), masks.to(dev), torch.zeros_like(ids, dtype=torch.bool).to(dev),
                          init_eta=eta.to(dev))
torch.set_num_threads(min(16, os.cpu_count() or 1))
total = None
for s in range(2):
    ad = VlmoRefAdapters(cpu_model, ids[s:s + 1], masks[s:s + 1])
    with torch.enable_grad():
        adv, losses = oracle.projected_gradient_descent(ad.pgd_attack, img[s:s + 1], 0.125, 0.01, STEPS, np.inf, clip_min=-1,
                                                        clip_max=1, y=ad.gen_ori_feats(img[s:s + 1]), ori_x=img[s:s + 1],
                                                        time=0, ls=1, flavor="vlmo", init_eta=eta[s:s + 1])
    got = res.adv_images[s].cpu()
    same = (got == adv[0]).float().mean().item()
    print(json.dumps(dict(sample=s, steps=STEPS, identical_pixels=round(same, 6),
                          max_abs_dev=round(float((got - adv[0]).abs().max()), 6),
                          mean_abs_dev=float((got - adv[0]).abs().mean()))), flush=True)
    total = np.array(losses) if total is None else total + np.array(losses)
print(json.dumps(dict(loss_gpu_batch=[round(v, 3) for v in res.loss_lists[0]], loss_cpu_sum=[round(float(v), 3) for v in total],
                      max_rel_loss_dev=float(np.max(np.abs(np.array(res.loss_lists[0]) - total) / np.abs(total))))))
