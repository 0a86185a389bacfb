#!/usr/bin/env python
"""ALBEF-flavor entry point in the style of the reference's ``ALBEF_attack/VQA.py`` (argparse + yaml, :119-134).

    python entry/VQA.py --config entry/configs/VQA.yaml [--output_dir out] [--seed 42] [--n_samples 128]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 entry/VQA.py --config ...
"""
import argparse
import os

import yaml

from _common import finish, init_distributed

import torch.distributed as dist  # noqa: E402  (after _common: it sets the HSA IPC mode before torch loads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(__file__), "configs", "VQA.yaml"))
    ap.add_argument("--output_dir", default="")
    ap.add_argument("--seed", default=42, type=int)
    ap.add_argument("--n_samples", default=None, type=int)
    ap.add_argument("--image_only", action="store_true", help="no word substitution (40-step image PGD)")
    ap.add_argument("--tiny", action="store_true", help="test-sized encoder")
    ap.add_argument("--dual_every", default=0, type=int,
                    help="every n-th synthetic sample's victim answer occurs in its paraphrase -> dual loss (old_alg == 0)")
    ap.add_argument("--mixed", action="store_true",
                    help="one bucket: batches mix schedules and loss modes (attack_mixed) instead of schedule-pure buckets")
    args = ap.parse_args()
    cfg = yaml.safe_load(open(args.config))
    rank, world, device = init_distributed()

    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base, albef_tiny
    mcfg = albef_tiny() if args.tiny else albef_base(image_size=cfg["image_res"])
    white = FrozenAlbef(mcfg, seed=args.seed).to(device)
    black = FrozenAlbef.finetuned_from(white, seed=args.seed + 1).to(device)
    out_dir = os.path.join(args.output_dir, cfg.get("attack_dir", "attack_dir")) if args.output_dir else None
    res = run_sweep("albef", white, black, AlbefAttackAdapters(white), args.n_samples or cfg["n_samples"],
                    cfg["batch_size_test"], mcfg.image_size, min(cfg["text_len"], 8 if args.tiny else 512), device,
                    rank, world, joint=not args.image_only, save_dir=out_dir, seed=args.seed,
                    max_words=4 if args.tiny else 12, dual_every=args.dual_every, mixed=args.mixed,
                    force_collective=dist.is_initialized())
    finish(rank, world, res, os.path.join(args.output_dir, "adv_txt.json") if args.output_dir else None)


if __name__ == "__main__":
    main()
