#!/usr/bin/env python
"""ALBEF-flavor entry point in the style of the reference's ``ALBEF_attack/VQA.py`` (argparse + yaml, :119-134).

    python entry/VQA.py --config entry/configs/VQA.yaml [--output_dir out] [--seed 42] [--n_samples 128]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 entry/VQA.py --config ...

File inputs instead of the synthetic set: ``--questions`` (default: the yaml's ``test_file``), ``--image_root``
(``vqa_root``), ``--vocab_file``, ``--tables_dir`` (``right_part*.txt``, ``albef_ans_table*.txt``, ``chatgpt_all_5k*.txt``
...), ``--checkpoint`` / ``--checkpoint_vqa`` (the reference's "pretrain model path" / "fine-tune model path",
``adv_attack.py:83,96``).  See ``vqattack_amd/attack/dataset.py``.
"""
import argparse
import os

import yaml

from _common import file_source, finish, init_distributed, load_checkpoint, mlm_proposer, seed_everything

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
    ap.add_argument("--questions", nargs="*", default=None, help="VQA annotation json file(s) (configs: test_file)")
    ap.add_argument("--image_root", default="", help="directory of the annotation's image paths (configs: vqa_root)")
    ap.add_argument("--vocab_file", default="", help="BERT vocab.txt (needed for textual questions / tables)")
    ap.add_argument("--tables_dir", default="", help="directory of the reference's in-tree *.txt tables")
    ap.add_argument("--checkpoint", default="", help="pre-trained ALBEF checkpoint -> white box (adv_attack.py:83)")
    ap.add_argument("--checkpoint_vqa", default="", help="VQA fine-tuned checkpoint -> victim (adv_attack.py:96)")
    ap.add_argument("--sim_threshold", default=0.95, type=float,
                    help="sentence-similarity floor of a substitution (adv_attack.py:303)")
    ap.add_argument("--mlm_checkpoint", default="", help="BertForMaskedLM state dict -> candidate proposer (adv_attack.py:110)")
    args = ap.parse_args()
    cfg = yaml.safe_load(open(args.config))
    rank, world, device = init_distributed()
    seed_everything(args.seed, rank)

    from vqattack_amd.attack.runner import AttackConfig
    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_base, albef_tiny
    from vqattack_amd.whitebox import checkpoint
    if args.checkpoint:
        white = checkpoint.albef_from_reference(load_checkpoint(args.checkpoint), image_size=cfg["image_res"],
                                                vqa_head=False).to(device)
        mcfg = white.cfg
        if args.checkpoint_vqa:
            black = checkpoint.albef_from_reference(load_checkpoint(args.checkpoint_vqa), image_size=cfg["image_res"],
                                                    vqa_head=True, strict=False).to(device)
        else:
            black = FrozenAlbef.finetuned_from(white, seed=args.seed + 1).to(device)
    else:
        mcfg = albef_tiny() if args.tiny else albef_base(image_size=cfg["image_res"])
        white = FrozenAlbef(mcfg, seed=args.seed).to(device)
        black = FrozenAlbef.finetuned_from(white, seed=args.seed + 1).to(device)
    out_dir = os.path.join(args.output_dir, cfg.get("attack_dir", "attack_dir")) if args.output_dir else None
    text_len = min(cfg["text_len"], 8 if args.tiny else 512)
    proposer, banned = mlm_proposer(args.mlm_checkpoint, args.vocab_file, device)
    source = None
    if args.questions:
        source = file_source("albef", args.questions, args.image_root or cfg.get("vqa_root", ""), text_len,
                             mcfg.image_size, args.vocab_file, args.tables_dir, joint=not args.image_only)
    res = run_sweep("albef", white, black, AlbefAttackAdapters(white), args.n_samples or cfg["n_samples"],
                    cfg["batch_size_test"], mcfg.image_size, text_len, device,
                    rank, world, joint=not args.image_only, save_dir=out_dir, seed=args.seed,
                    max_words=4 if args.tiny else 12, dual_every=args.dual_every, mixed=args.mixed,
                    force_collective=dist.is_initialized(), source=source, mlm_logits_fn=proposer, banned_ids=banned,
                    config=AttackConfig(sim_threshold=args.sim_threshold))
    finish(rank, world, res, os.path.join(args.output_dir, "adv_txt.json") if args.output_dir else None)


if __name__ == "__main__":
    main()
