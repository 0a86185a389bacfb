"""Eager vs hipGraph-replayed PGD at the reference's own batch size (1) on VLMO-base: ms per PGD iteration."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqattack_amd  # noqa: E402
from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_base  # noqa: E402

dev = "cuda:0"
cfg = vlmo_base(384)
model = FrozenVlmo(cfg, seed=0).to(dev)
ad = VlmoAttackAdapters(model)
fused_modes = [True, False] if os.environ.get("AB_FUSED", "1") == "1" else [True]
for fused, batch in [(f, b) for f in fused_modes for b in (1, 4, 16)]:
    model.fused_blocks = fused          # False: the eager nn.Module block loop (autograd), round 3's path
    ids = torch.zeros(batch, 40, dtype=torch.long, device=dev)
    ids[:, 0], ids[:, 1:9], ids[:, 9] = 101, 2054, 102
    ad.set_text(ids, (ids != 0).long())
    x0 = torch.empty(batch, 3, 384, 384, device=dev).uniform_(-1, 1)
    y = ad.gen_ori_feats(x0)
    for graph in (False, True):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.enable_grad():
                vqattack_amd.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, 40, np.inf, clip_min=-1, clip_max=1,
                                                        y=list(y), ori_x=x0, time=0, ls=1, flavor="vlmo", graph=graph,
                                                        sanity_checks=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%-14s batch %2d  graph=%-5s  %.2f ms/iteration  %.2f examples/s" % (
            "fused blocks" if fused else "eager blocks", batch, graph, dt / 40 * 1e3, batch / dt), flush=True)
