"""Why does a success bit of tests/test_success_bits_base.py differ?  For the named samples of one fixture: the product's
attack (the same batch slice the test runs) and the ORACLE's adversarial image (cached by tests/golden/make_asr_fixture.py
in the build container; copy the sample's ``sNNNN.npz`` under ``tools/_probe_inputs/<flavor>_seed<seed>_b<budget>/`` --
that directory travels to the GPU box, ``gpurun_out/`` does not) are scored by the victim on the GPU AND on the CPU:
if the four decisions split by IMAGE and not by SCORER, and the victim's two leading answers are within the 1e-3 tie
margin on both images, the bit is decided by which of two equally adversarial images is scored -- the case
tests/test_success_bits_base.py tolerates -- and not by a defect of the attack or of the scorer.

    python tools/asr_tie_probe.py --fixture tests/golden/asr_base_albef_s29.json --samples 35 >> profiles/r05/asr_tie_probes.jsonl

Test infrastructure (imports ``oracle/`` for the CPU scorer); reference: adv_attack.py:717-733, vlmo_module.py:2063-2091.
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
from tests import test_success_bits as tsb  # noqa: E402


def top2(values, ids=None):
    v, i = values.topk(2)
    pick = i if ids is None else ids[i]
    return dict(answers=[int(x) for x in pick], scores=[round(float(x), 6) for x in v], margin=round(float(v[0] - v[1]), 6))


def score(flavor, black, image, ids, masks, k):
    """Leading two answers of the victim (device-agnostic: runs wherever ``black`` lives)."""
    with torch.no_grad():
        if flavor == "vlmo":
            _, states = black.encode(image, black.text_embeddings(ids), masks)
            return top2(black.vqa_classifier(black.pooled(states))[0, :k])
        n = int(masks[0].sum())
        st, _ = black.visual_encoder(image)
        q, _ = black.text_encoder(black.text_embeddings(ids[:, :n]), masks[:, :n], st)
        if image.is_cuda:
            ti, tp = black.rank_answer(q, masks[:, :n])
        else:
            from oracle import blackbox_ref as bb
            ans = black.answer_ids
            ti, tp = bb.rank_answer(black._decode, q, masks[:, :n], ans, (ans != black.cfg.pad_id).long(),
                                    min(black.cfg.k_test, black.cfg.n_answers), pad_id=black.cfg.pad_id)
        return dict(answers=[int(x) for x in ti[0, :2]], scores=[round(float(x), 6) for x in tp[0, :2]],
                    margin=round(float(tp[0, 0] - tp[0, 1]), 6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", required=True)
    ap.add_argument("--samples", required=True, help="comma list of sample indices")
    ap.add_argument("--oracle-adv", default=None, help="directory with the oracle's sNNNN.npz (default: tools/_probe_inputs/...)")
    args = ap.parse_args()
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    rec = json.load(open(args.fixture))
    flavor, n, k = rec["flavor"], rec["n"], rec["n_answers"]
    cache = args.oracle_adv or os.path.join(ROOT, "tools", "_probe_inputs",
                                            "{}_seed{}_b{}".format(flavor, rec["seed"], rec["budget"]))
    dev = torch.device("cuda", 0)
    cfg_kw = dict(n_answers=k, k_test=min(128, k)) if flavor == "albef" else {}
    white, black, adapters_cls, _, cfg = tsb.build(flavor, "base", **cfg_kw)
    shape = rec["shape"]
    ids, masks, att, tasks, _, images, eta = tsb.make_samples(flavor, cfg, n=n, seed=rec["seed"],
                                                             words=tuple(shape["words"]), max_att=shape["max_att"],
                                                             text_len=shape["text_len"])
    proposals = [[(int(p), [int(v) for v in vs]) for p, vs in row] for row in rec["proposals"]]
    white_gpu, black_gpu = copy.deepcopy(white).to(dev), copy.deepcopy(black).to(dev)
    attack = BatchedVQAttack(adapters_cls(white_gpu), flavor, white_gpu.embedding_tables(),
                             AttackConfig(budget=rec["budget"], sanity_checks=True, sim_threshold=rec["sim_threshold"]),
                             similarity_fn=text_update.BagOfEmbeddingsSimilarity(seed=rec["sim_seed"]))
    batch = 50 if flavor == "vlmo" else 32              # the slices of tests/test_success_bits_base.py (full mode)
    for s in [int(v) for v in args.samples.split(",")]:
        lo = s // batch * batch
        sl = slice(lo, min(lo + batch, rec.get("n_scored", n)))      # the test's slice (a partial set ends at n_scored)
        res = attack.attack_mixed(images[sl].to(dev), ids[sl].to(dev), masks[sl].to(dev), att[sl].to(dev),
                                  init_eta=eta[sl].to(dev), proposals=proposals[sl], tasks=tasks[sl])
        prod_img, prod_ids = res.adv_images[s - lo:s - lo + 1].cpu(), res.adv_text_ids[s - lo:s - lo + 1].cpu()
        z = np.load(os.path.join(cache, "s{:04d}.npz".format(s)))
        ora_img = torch.from_numpy(z["adv"])[None]
        ora_ids = ids[s:s + 1].clone()
        ora_ids[0, :len(z["ids"])] = torch.from_numpy(z["ids"])
        m = masks[s:s + 1]
        out = dict(fixture=os.path.basename(args.fixture), flavor=flavor, sample=s, n_answers=k,
                   recorded_oracle_bit=rec["success_bits"][s], recorded_oracle_margin=rec["adversarial_margins"][s],
                   clean_answer=rec["clean_answers"][s],
                   pixels_bit_identical=round(float((prod_img == ora_img).float().mean()), 6),
                   max_abs_pixel_diff=round(float((prod_img - ora_img).abs().max()), 4),
                   substituted_ids_equal=bool(torch.equal(prod_ids[:, :ora_ids.shape[1]], ora_ids)))
        for name, img, tid in (("product_image", prod_img, prod_ids), ("oracle_image", ora_img, ora_ids)):
            out[name] = dict(gpu_scorer=score(flavor, black_gpu, img.to(dev), tid.to(dev), m.to(dev), k),
                             cpu_scorer=score(flavor, black, img, tid, m, k))
        a = [out[i][j]["answers"][0] for i in ("product_image", "oracle_image") for j in ("gpu_scorer", "cpu_scorer")]
        out["decision_follows"] = ("the image (both scorers agree on either image)" if a[0] == a[1] and a[2] == a[3]
                                   else "the scorer")
        out["tied_on_both_images"] = all(abs(out[i][j]["margin"]) < 1e-3 for i in ("product_image", "oracle_image")
                                         for j in ("gpu_scorer", "cpu_scorer"))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
