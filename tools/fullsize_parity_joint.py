"""Full-size parity of the JOINT attack (image PGD blocks + text-gradient probes + word substitution) and of the ALBEF
flavor: shipped batched HIP path on the MI355X vs the per-sample CPU oracle (oracle/attack_loop.py).

    FLAVOR=vlmo|albef WORDS=3 BUDGET=40 python tools/fullsize_parity_joint.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import attack_loop  # noqa: E402
from oracle.adapters_ref import AlbefRefAdapters, VlmoRefAdapters  # noqa: E402
from vqattack_amd.attack import text_update  # noqa: E402
from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack  # noqa: E402

FLAVOR = os.environ.get("FLAVOR", "vlmo")
WORDS = int(os.environ.get("WORDS", "3"))
BUDGET = int(os.environ.get("BUDGET", "40"))
dev = torch.device("cuda", 0)
if FLAVOR == "vlmo":
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base
    cfg = vlmo_base(384)
    gpu_model, cpu_model = FrozenVlmo(cfg, seed=0).to(dev), FrozenVlmo(cfg, seed=0)
    adapters, ref_cls, text_len = VlmoAttackAdapters(gpu_model), VlmoRefAdapters, 40
else:
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base
    cfg = albef_base(384, mlm_probability=0.0)
    gpu_model, cpu_model = FrozenAlbef(cfg, seed=0).to(dev), FrozenAlbef(cfg, seed=0)
    adapters, ref_cls, text_len = AlbefAttackAdapters(gpu_model), AlbefRefAdapters, 12
ids = torch.zeros(2, text_len, dtype=torch.long)
ids[0, :7] = torch.tensor([101, 2054, 3609, 2003, 1996, 4937, 102])
ids[1, :9] = torch.tensor([101, 2129, 2116, 6077, 2024, 1999, 1996, 2833, 102])
masks = (ids != 0).long()
att = torch.zeros_like(ids, dtype=torch.bool)
if WORDS:
    att[:, 1:1 + WORDS] = True
g = torch.Generator().manual_seed(0)
img = torch.empty(2, 3, 384, 384).uniform_(-1, 1, generator=g)
eta = torch.empty_like(img).uniform_(-0.125, 0.125, generator=g)
sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
attack = BatchedVQAttack(adapters, FLAVOR, gpu_model.embedding_tables(),
                         AttackConfig(budget=BUDGET, sanity_checks=True, sim_threshold=0.3), similarity_fn=sim)
proposals = None
if WORDS:
    proposals = text_update.propose_candidates(adapters.mlm_logits(ids.to(dev), masks.to(dev)), ids, att, threshold=0)
res = attack.attack_batch(img.to(dev), ids.to(dev), masks.to(dev), att.to(dev), init_eta=eta.to(dev), proposals=proposals)
torch.set_num_threads(min(16, os.cpu_count() or 1))
for s in range(2):
    n = int(masks[s].sum()) if FLAVOR == "albef" else text_len
    adv, new_ids, losses = attack_loop.attack_one(ref_cls, cpu_model, FLAVOR, img[s:s + 1], ids[s:s + 1, :n], masks[s:s + 1, :n],
                                                  proposals[s] if WORDS else None, sim, init_eta=eta[s:s + 1],
                                                  budget=BUDGET, sim_threshold=0.3)
    got = res.adv_images[s].cpu()
    print(json.dumps(dict(flavor=FLAVOR, sample=s, words=WORDS, gradient_steps=res.gradient_steps,
                          identical_pixels=round((got == adv[0]).float().mean().item(), 6),
                          max_abs_dev=round(float((got - adv[0]).abs().max()), 6),
                          text_ids_equal=res.adv_text_ids[s, :n].cpu().tolist() == new_ids[0].tolist(),
                          substituted=int((res.adv_text_ids[s].cpu() != ids[s]).sum()))), flush=True)
