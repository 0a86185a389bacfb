#!/usr/bin/env python
"""VLMO-flavor entry point in the style of the reference's sacred CLI (``VLMO_VQAttack/run.py:69-185``):

    python entry/run.py with task_finetune_vqa_base_image384 test_only=True per_gpu_batchsize=64 n_samples=128

Named configs and ``key=value`` overrides are parsed like sacred's ``with`` clause; only the keys the attack reads are
known (``vlmo/config.py:20-90,283-337``): image_size, max_text_len, per_gpu_batchsize, model arch, seed, test_only.
"""
import os
import sys

from _common import finish, init_distributed

import torch.distributed as dist  # noqa: E402  (after _common: it sets the HSA IPC mode before torch loads)

NAMED = {
    "task_finetune_vqa_base_image384": dict(arch="vlmo_base", image_size=384),
    "task_finetune_vqa_base_image480": dict(arch="vlmo_base", image_size=480),
    "task_finetune_vqa_large_image384": dict(arch="vlmo_large", image_size=384),
    "task_finetune_vqa_large_image480": dict(arch="vlmo_large", image_size=480),
    "tiny": dict(arch="vlmo_tiny", image_size=32, max_text_len=8),
}
DEFAULTS = dict(arch="vlmo_base", image_size=384, max_text_len=40, per_gpu_batchsize=64, seed=1, test_only=True,
                n_samples=128, image_only=False, attack_dir="", dual_every=0, mixed=False)


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
    from vqattack_amd.attack.sweep import run_sweep
    from vqattack_amd.whitebox import vlmo
    mcfg = vlmo.vlmo_tiny() if cfg["arch"] == "vlmo_tiny" else getattr(vlmo, cfg["arch"])(
        image_size=cfg["image_size"], max_text_len=cfg["max_text_len"])
    white = vlmo.FrozenVlmo(mcfg, seed=cfg["seed"]).to(device)
    black = vlmo.FrozenVlmo.finetuned_from(white, seed=cfg["seed"] + 1).to(device)
    res = run_sweep("vlmo", white, black, vlmo.VlmoAttackAdapters(white), cfg["n_samples"], cfg["per_gpu_batchsize"],
                    mcfg.image_size, mcfg.max_text_len, device, rank, world, joint=not cfg["image_only"],
                    save_dir=cfg["attack_dir"] or None, seed=cfg["seed"],
                    max_words=4 if cfg["arch"] == "vlmo_tiny" else 12, dual_every=cfg["dual_every"], mixed=cfg["mixed"],
                    force_collective=dist.is_initialized())
    # adversarial images <qid>.pt and the adversarial-text json go to attack_dir (vlmo_module.py:166-167,2059-2062,2095-2097)
    finish(rank, world, res, os.path.join(cfg["attack_dir"], "adv_txt.json") if cfg["attack_dir"] else None)


if __name__ == "__main__":
    main()
