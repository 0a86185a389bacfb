#!/usr/bin/env python
"""VLMO-flavor entry point in the style of the reference's sacred CLI (``VLMO_VQAttack/run.py:69-185``):

    python entry/run.py with task_finetune_vqa_base_image384 test_only=True per_gpu_batchsize=64 n_samples=128

Named configs and ``key=value`` overrides are parsed like sacred's ``with`` clause; only the keys the attack reads are
known (``vlmo/config.py:20-90,283-337``): image_size, max_text_len, per_gpu_batchsize, model arch, seed, test_only, and
the two checkpoint paths ``pretrain_path`` (white box) / ``load_path`` (VQA victim) (``config.py:83-87``);
``mlm_checkpoint=`` is a local ``BertForMaskedLM`` state dict for the candidate proposer (``adv_attack.py:110``).

File inputs instead of the synthetic set (``attack/dataset.py``): ``questions=`` (VQA annotation json), ``image_root=``,
``vocab_file=`` (BERT vocab.txt, for textual questions / tables), ``tables_dir=`` (the reference's in-tree ``*.txt``
tables: ``right_part``, ``vlmo_ans_table``, ``vilt_ans_table_for_chatgpt``, ``chatgpt_all_5k``, ``all_correct_ans``).
8-bit images are resized + normalised on the device (Pillow-exact), the next batch's files are read while the current
one is attacked; ``attack_dir=`` receives ``<question_id>.pt`` and ``adv_txt.json``.
"""
import os
import sys

from _common import file_source, finish, init_distributed, load_checkpoint, mlm_proposer, seed_everything

import torch.distributed as dist  # noqa: E402  (after _common: it sets the HSA IPC mode before torch loads)

NAMED = {
    "task_finetune_vqa_base_image384": dict(arch="vlmo_base", image_size=384),
    "task_finetune_vqa_base_image480": dict(arch="vlmo_base", image_size=480),
    "task_finetune_vqa_large_image384": dict(arch="vlmo_large", image_size=384),
    "task_finetune_vqa_large_image480": dict(arch="vlmo_large", image_size=480),
    "tiny": dict(arch="vlmo_tiny", image_size=32, max_text_len=8),
}
DEFAULTS = dict(arch="vlmo_base", image_size=384, max_text_len=40, per_gpu_batchsize=64, seed=1, test_only=True,
                n_samples=128, image_only=False, attack_dir="", dual_every=0, mixed=False, questions="", image_root="",
                vocab_file="", tables_dir="", pretrain_path="", load_path="", mlm_checkpoint="", sim_threshold=0.95)


def parse(argv):
    cfg = dict(DEFAULTS)
    if argv and argv[0] == "with":
        argv = argv[1:]
    for tok in argv:
        if "=" in tok:
            k, v = tok.split("=", 1)
            if k not in cfg:
                raise SystemExit("unknown config key '{}'".format(k))
            old = cfg[k]
            cfg[k] = (v.lower() in ("1", "true", "yes")) if isinstance(old, bool) else type(old)(v)
        elif tok in NAMED:
            cfg.update(NAMED[tok])
        else:
            raise SystemExit("unknown named config '{}' (known: {})".format(tok, ", ".join(NAMED)))
    return cfg


def main():
    cfg = parse(sys.argv[1:])
    if not cfg["test_only"]:
        raise SystemExit("only the attack (test_only=True) is implemented; training is out of scope")
    rank, world, device = init_distributed()
    seed_everything(cfg["seed"], rank)
    from vqattack_amd.attack.runner import AttackConfig
    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox import vlmo
    from vqattack_amd.whitebox import checkpoint
    if cfg["pretrain_path"]:
        # the reference's two checkpoints: pre-trained = white box, VQA fine-tuned = victim (vlmo_module.py:303-324,330-567)
        white = checkpoint.vlmo_from_reference(load_checkpoint(cfg["pretrain_path"]), image_size=cfg["image_size"],
                                               max_text_len=cfg["max_text_len"], vqa_head=False).to(device)
        mcfg = white.cfg
        if cfg["load_path"]:
            black = checkpoint.vlmo_from_reference(load_checkpoint(cfg["load_path"]), image_size=cfg["image_size"],
                                                   max_text_len=cfg["max_text_len"], vqa_head=True, strict=False).to(device)
        else:
            black = vlmo.FrozenVlmo.finetuned_from(white, seed=cfg["seed"] + 1).to(device)
    else:
        mcfg = vlmo.vlmo_tiny() if cfg["arch"] == "vlmo_tiny" else getattr(vlmo, cfg["arch"])(
            image_size=cfg["image_size"], max_text_len=cfg["max_text_len"])
        white = vlmo.FrozenVlmo(mcfg, seed=cfg["seed"]).to(device)
        black = vlmo.FrozenVlmo.finetuned_from(white, seed=cfg["seed"] + 1).to(device)
    proposer, banned = mlm_proposer(cfg["mlm_checkpoint"], cfg["vocab_file"], device)   # HF bert-base-uncased MLM
    source = None
    if cfg["questions"]:
        source = file_source("vlmo", cfg["questions"], cfg["image_root"], mcfg.max_text_len, mcfg.image_size,
                             cfg["vocab_file"], cfg["tables_dir"], joint=not cfg["image_only"])
    res = run_sweep("vlmo", white, black, vlmo.VlmoAttackAdapters(white), cfg["n_samples"], cfg["per_gpu_batchsize"],
                    mcfg.image_size, mcfg.max_text_len, device, rank, world, joint=not cfg["image_only"],
                    save_dir=cfg["attack_dir"] or None, seed=cfg["seed"],
                    max_words=4 if cfg["arch"] == "vlmo_tiny" else 12, dual_every=cfg["dual_every"], mixed=cfg["mixed"],
                    force_collective=dist.is_initialized(), source=source, mlm_logits_fn=proposer, banned_ids=banned,
                    config=AttackConfig(sim_threshold=cfg["sim_threshold"]))      # adv_attack.py:303: 0.95
    # adversarial images <qid>.pt and the adversarial-text json go to attack_dir (vlmo_module.py:166-167,2059-2062,2095-2097)
    finish(rank, world, res, os.path.join(cfg["attack_dir"], "adv_txt.json") if cfg["attack_dir"] else None)


if __name__ == "__main__":
    main()
