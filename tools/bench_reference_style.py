"""The reference's OWN call shape, alone: batch 1, its packed ``pgd_attack`` closure through the drop-in PGD (what
``bench.py``'s ``reference_style`` leg times), for profiling (``rocprofv3 --kernel-trace --stats -- python3 tools/...``)
and for recording library GEMM solutions of the batch-1 shapes (TunableOp tuning ON: ``tools/tune_gemms.sh``'s
``batch1`` set).  One JSON line: ms per PGD iteration for eager / hipGraph-replayed iterations, both flavors' base models
at 384 px (and VLMO-base at 480 px with ``--image480``: 915 / 901-token shapes).
usage: python3 tools/bench_reference_style.py [--models vlmo_base,albef_base] [--image480] [--steps 40] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqattack_amd import dropin  # noqa: E402
from vqattack_amd.whitebox import reference_style, tuned_gemms  # noqa: E402


def build(model, image_size, dev):
    if model == "vlmo_base":
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, vlmo_base
        cfg = vlmo_base(image_size)
        return "vlmo", FrozenVlmo(cfg, seed=0).to(dev), cfg, cfg.max_text_len
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_base
    cfg = albef_base(image_size)
    return "albef", FrozenAlbef(cfg, seed=0).to(dev), cfg, 40


def question(text_len, n_words, dev):
    ids = torch.zeros(1, text_len, dtype=torch.long, device=dev)
    ids[0, 0], ids[0, 1 + n_words] = 101, 102
    ids[0, 1:1 + n_words] = torch.arange(2000, 2000 + n_words, device=dev)
    return ids, (ids != 0).long()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="vlmo_base")
    ap.add_argument("--image480", action="store_true")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--words", type=int, default=12, help="question words: 12 -> 14 real tokens (591-token layout)")
    ap.add_argument("--graph", default="both", choices=["both", "off", "on"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tuned = tuned_gemms.enable() if os.environ.get("PYTORCH_TUNABLEOP_TUNING", "0") != "1" else False
    out = {"tuned_gemms": bool(tuned), "steps": args.steps, "runs": []}
    for model in args.models.split(","):
        for size in ([384, 480] if args.image480 else [384]):
            flavor, white, cfg, text_len = build(model, size, dev)
            ids, masks = question(text_len, args.words, dev)
            image = torch.empty(1, 3, size, size, device=dev).uniform_(-1, 1)
            pgd = dropin.load(flavor).projected_gradient_descent.projected_gradient_descent
            batch = dict(text_ids=ids, text_masks=masks)
            if flavor == "vlmo":
                me = reference_style.VlmoReferenceClosures(white, batch)
                y = me.Gen_ori_feats(image)
            else:
                me = reference_style.AlbefReferenceClosures(white, batch)
                img_feats, txt_feats = me.Gen_ori_feats(image)
                y = [txt_feats, img_feats, None, None, None]
            for graph in ([False, True] if args.graph == "both" else [args.graph == "on"]):
                kw = dict(graph=True) if graph else {}

                def call():
                    with torch.enable_grad():
                        return pgd(me.pgd_attack, image, 0.125, 0.01, args.steps, np.inf, -1, 1, y=list(y), time=0,
                                   ori_x=image, ls=1, **kw)
                call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    call()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.reps / args.steps * 1e3
                out["runs"].append(dict(model=model, image_size=size, graph=graph, ms_per_pgd_iteration=round(ms, 3)))
                print("# {} {} px graph={} {:.3f} ms / PGD iteration".format(model, size, graph, ms), file=sys.stderr,
                      flush=True)
            del white, me
            torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
