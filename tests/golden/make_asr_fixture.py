#!/usr/bin/env python
"""Generate tests/golden/asr_base_<flavor>.json: the ORACLE side of the attack-success-bit comparison at the model size
BASELINE.json names (VLMO-base / ALBEF-base, 384 px, the full 40-step budget), computed in the BUILD container on CPU.

    python tests/golden/make_asr_fixture.py --flavor vlmo  --n 200
    python tests/golden/make_asr_fixture.py --flavor albef --n 64
    python tests/golden/make_asr_fixture.py --flavor vlmo  --n 300 --seed 29 --out tests/golden/asr_base_vlmo_s29.json
    python tests/golden/make_asr_fixture.py --flavor albef --n 128 --seed 29 --sizes 8,12,16 \
        --out tests/golden/asr_base_albef_s29.json
    python tests/golden/make_asr_fixture.py --flavor albef --n 96 --seed 31 --sizes 8,12,16 \
        --out tests/golden/asr_base_albef_s31.json

What is stored is data only: the seeds and shape parameters that regenerate the inputs (tests/test_success_bits.py
``make_samples``), the candidate proposals (host data injected on both sides), and the oracle pipeline's outputs -- the
victim's clean answers, its answers to the adversarial pairs, the substituted token ids, the success bits and the
decision margins.  tests/test_success_bits_base.py (-m gpu) runs ONLY the product on the same inputs and compares.

Oracle pipeline (test infrastructure): oracle/attack_loop.attack_one (per-sample block loop, batch 1, pinned by the
reference's own loop code) + oracle/blackbox_ref (per-question scorers); reference: ``adv_attack.py:559-733``,
``vlmo_module.py:1892-2091``.

The synthetic victim with its full answer vocabulary flips on (almost) every perturbed pair at this depth (round 3: ASR
1.0 on 24 samples), which makes equal bits uninformative.  A question of the reference's data set has a CLOSED answer
set far smaller than the vocabulary (yes / no, a number, a colour, ...), so the victim here answers from a closed set
of ``n_answers`` classes chosen such that the oracle's ASR on the sample set lands inside 0.3 .. 0.7 (the attack itself
never sees the victim, so one attack run serves every candidate size).  Adversarial images are cached under
``--cache`` (not tracked, 1.8 MB each) so that an interrupted run resumes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_success_bits as tsb  # noqa: E402

SHAPE = dict(vlmo=dict(words=(4, 13), max_att=5, text_len=None),      # n ~ U{4..12} words
             albef=dict(words=(4, 13), max_att=5, text_len=16))
CANDIDATE_SIZES = [2, 3, 4, 6, 8, 16, 64, 3129]


def log(msg):
    print("[asr_fixture] " + msg, file=sys.stderr, flush=True)


def proposals_on_cpu(adapters, ids, masks, att, chunk=8):
    from vqattack_amd.attack import text_update
    out = []
    for lo in range(0, ids.shape[0], chunk):
        sl = slice(lo, lo + chunk)
        out += text_update.propose_candidates(adapters.mlm_logits(ids[sl], masks[sl]), ids[sl], att[sl], threshold=0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flavor", required=True, choices=["vlmo", "albef"])
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--budget", type=int, default=40)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cache", default=os.path.join(ROOT, "gpurun_out", "asr_cache"))   # gpurun_out/ never travels to the GPU box
    ap.add_argument("--out", default=None)
    ap.add_argument("--upto", type=int, default=None,
                    help="attack and score only the first UPTO samples of the --n sample set (the inputs are still those of "
                         "the n-sample draw: the random start of sample s depends on n); the fixture records n_scored. "
                         "Turns an interrupted run's cache into a usable fixture")
    ap.add_argument("--time-budget", type=float, default=None,
                    help="seconds: stop attacking further samples after this long and write the fixture for the prefix "
                         "done so far (n_scored), so that a run with a hard wall-clock limit always leaves a usable file")
    ap.add_argument("--sizes", default=None,
                    help="comma list of closed-answer-set sizes to score (default: CANDIDATE_SIZES); the attack results are "
                         "cached, so re-scoring with other sizes costs minutes")
    args = ap.parse_args()
    sizes = [int(v) for v in args.sizes.split(",")] if args.sizes else CANDIDATE_SIZES
    torch.set_num_threads(args.threads)
    from oracle import attack_loop
    from vqattack_amd.attack import text_update
    flavor = args.flavor
    out_path = args.out or os.path.join(ROOT, "tests", "golden", "asr_base_{}.json".format(flavor))
    cache = os.path.join(args.cache, "{}_seed{}_b{}".format(flavor, args.seed, args.budget))
    os.makedirs(cache, exist_ok=True)

    white, black, adapters_cls, ref_cls, cfg = tsb.build(flavor, "base")
    shape = SHAPE[flavor]
    samples = tsb.make_samples(flavor, cfg, n=args.n, seed=args.seed, **shape)
    ids, masks, att, _tasks, oracle_tasks, images, eta = samples
    n_all = args.n
    if args.upto is not None:            # a prefix of the set: everything below runs over the first `upto` samples
        args.n = min(args.upto, n_all)
        ids, masks, att, images, eta = ids[:args.n], masks[:args.n], att[:args.n], images[:args.n], eta[:args.n]
        oracle_tasks = oracle_tasks[:args.n]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=5)
    prop_path = os.path.join(cache, "proposals_n{}.json".format(n_all))
    if os.path.exists(prop_path):
        proposals = [[(int(p), [int(v) for v in vs]) for p, vs in row] for row in json.load(open(prop_path))]
    else:
        log("candidate proposals on the CPU ...")
        full = tsb.make_samples(flavor, cfg, n=n_all, seed=args.seed, **shape)
        proposals = proposals_on_cpu(adapters_cls(white), full[0], full[1], full[2])
        json.dump(proposals, open(prop_path, "w"))
    proposals = proposals[:args.n]

    # ---- the attack (never sees the victim): once per sample, cached
    t0 = time.perf_counter()
    adv_ids = ids.clone()
    last_losses = []
    with tsb.oracle_text_len(flavor, ids.shape[1]):
        for s in range(args.n):
            if args.time_budget and time.perf_counter() - t0 > args.time_budget:
                log("time budget reached after {} samples".format(s))
                if s == 0:
                    raise SystemExit("--time-budget {} s was reached before a single sample had been attacked: nothing "
                                     "to score, no fixture written".format(args.time_budget))
                args.n = s
                ids, masks, att, images, eta = ids[:s], masks[:s], att[:s], images[:s], eta[:s]
                oracle_tasks, proposals, adv_ids = oracle_tasks[:s], proposals[:s], adv_ids[:s]
                break
            path = os.path.join(cache, "s{:04d}.npz".format(s))
            n = int(masks[s].sum()) if flavor == "albef" else ids.shape[1]
            if os.path.exists(path):
                z = np.load(path)
                adv_ids[s, :n] = torch.from_numpy(z["ids"])
                last_losses.append(float(z["loss"]))
                continue
            adv, new_ids, losses = attack_loop.attack_one(
                ref_cls, white, flavor, images[s:s + 1], ids[s:s + 1, :n], masks[s:s + 1, :n],
                proposals[s] if proposals[s] else None, sim, init_eta=eta[s:s + 1], budget=args.budget,
                sim_threshold=0.3, task=oracle_tasks[s])
            np.savez(path, adv=adv.detach().numpy()[0], ids=new_ids[0].numpy(), loss=np.float32(losses[-1][-1]))
            adv_ids[s, :n] = new_ids[0]
            last_losses.append(float(losses[-1][-1]))
            log("{} sample {}/{} attacked ({} words, {}), {:.0f} s".format(
                flavor, s + 1, args.n, len(proposals[s]), "dual" if oracle_tasks[s] is not None else "feature",
                time.perf_counter() - t0))

    def adv_image(s):
        return torch.from_numpy(np.load(os.path.join(cache, "s{:04d}.npz".format(s)))["adv"])[None]

    # ---- the victim: clean and adversarial decisions for every candidate size of its closed answer set
    log("scoring ...")
    table = {}
    if flavor == "vlmo":
        from oracle import blackbox_ref as bb
        logits_clean, logits_adv = [], []
        with torch.no_grad():
            for s in range(args.n):
                for img, tid, store in ((images[s:s + 1], ids[s:s + 1], logits_clean),
                                        (adv_image(s), adv_ids[s:s + 1], logits_adv)):
                    _, states = black.encode(img, black.text_embeddings(tid), masks[s:s + 1])
                    store.append(black.vqa_classifier(black.pooled(states))[0])
        lc, la = torch.stack(logits_clean), torch.stack(logits_adv)
        for k in sizes:
            clean, after = bb.vlmo_predict(lc[:, :k]), bb.vlmo_predict(la[:, :k])
            top = la[:, :k].topk(2).values
            table[k] = (clean, after, (top[:, 0] - top[:, 1]).tolist())
    else:
        for k in sizes:
            if k > 128:
                continue
            _, black_k, _, _, _ = tsb.build(flavor, "base", n_answers=k, k_test=min(128, k))
            clean, _ = tsb.oracle_answers(flavor, black_k, images, ids, masks)
            after, margins = [], []
            for s in range(args.n):
                a, m = tsb.oracle_answers(flavor, black_k, adv_image(s), adv_ids[s:s + 1], masks[s:s + 1])
                after += a
                margins += m
            table[k] = (clean, after, margins)
            log("albef victim with {} answers: ASR {:.3f}".format(k, float(np.mean([int(a != c) for a, c in
                                                                                     zip(after, clean)]))))
    asr = {k: float(np.mean([int(a != c) for a, c in zip(v[1], v[0])])) for k, v in table.items()}
    log("oracle ASR by answer-set size: " + json.dumps(asr))
    inside = [k for k in table if 0.3 <= asr[k] <= 0.7]
    pick = max(inside) if inside else min(table, key=lambda k: abs(asr[k] - 0.5))
    clean, after, margins = table[pick]
    bits = [int(a != c) for a, c in zip(after, clean)]
    rec = dict(
        note="generated by tests/golden/make_asr_fixture.py in the build container (CPU oracle); data only.  The scored "
             "prefix n_scored (and with it the outcome-dependent choice of n_answers) depends on the host's speed when "
             "--time-budget cut the run: regenerate THIS file with  --n {} --seed {} --upto {}  (no --time-budget)".format(
                 n_all, args.seed, args.n),
        regenerate=dict(n=n_all, seed=args.seed, upto=args.n, sizes=[int(k) for k in sizes], budget=args.budget),
        flavor=flavor, size="base", n=n_all, n_scored=args.n, seed=args.seed, budget=args.budget, sim_threshold=0.3, sim_seed=5,
        white_seed=3, black_seed=4, shape=dict(words=list(shape["words"]), max_att=shape["max_att"],
                                               text_len=shape["text_len"]),
        n_answers=pick, oracle_asr=asr[pick], oracle_asr_by_answer_set_size={str(k): v for k, v in asr.items()},
        dual_loss_samples=sum(t is not None for t in oracle_tasks),
        proposals=proposals, clean_answers=clean, adversarial_answers=after, success_bits=bits,
        adversarial_margins=[round(float(m), 6) for m in margins], adv_text_ids=adv_ids.tolist(),
        final_losses=[round(x, 4) for x in last_losses])
    json.dump(rec, open(out_path, "w"))
    log("wrote {} (n_answers {}, oracle ASR {:.3f})".format(out_path, pick, asr[pick]))


if __name__ == "__main__":
    main()
