"""Attack-success bits at the model size BASELINE.json names, against the ORACLE pipeline's recorded outputs.

``tests/golden/asr_base_<flavor>[_<tag>].json`` hold what the CPU oracle (``oracle/attack_loop`` per-sample loop with the
full 40-step budget + ``oracle/blackbox_ref`` per-question scorers; reference: ``adv_attack.py:559-733``,
``vlmo_module.py:1892-2091``) produced in the build container for seeded samples at VLMO-base / ALBEF-base size, 384 px
(``tests/golden/make_asr_fixture.py``; questions of 4..12 words with 0..4 substitutable words, every 4th sample
dual-loss; one file per independent draw: other images, questions, schedules): the victim's clean answers, its answers
to the adversarial pairs, the substituted token ids, the success bits and the decision margins.  Here ONLY the product
runs -- the batched joint attack on the MI355X (``attack_mixed``: HIP operators, mixed schedules and loss modes in one
batch) and the batched black-box scorer -- on the same regenerated inputs, so no GPU time is spent waiting for the CPU.

Two modes.  The driver-run suite (default) attacks a FIXED SUBSET -- the first ``SUBSET[flavor]`` samples of every file
(``SUBSET_SMALL[flavor]`` of a draw with fewer than 100 samples), no selection by outcome -- to stay inside the suite's
time budget.  ``VQA_ASR_FULL=1`` attacks every sample of every
file (the pool the north star's "+-0.5 % on the same 5k pairs" is judged on; run once per round through gpurun, the
per-sample report written to ``$VQA_ASR_REPORT`` is tracked under ``profiles/``).

The victim answers from a closed answer set (``n_answers`` in the fixture) chosen so that the oracle's attack success
rate lies inside 0.3 .. 0.7: with the full answer vocabulary the synthetic victim flips on every perturbed pair (round 3:
ASR 1.0), and equal bits would say nothing.  Required (north_star: ASR within +-0.5 % of the reference's): clean answers
equal, substituted ids equal, success bits equal except where the victim is tied between its two leading answers
(oracle decision margin < 1e-3; at most 0.5 % of a set, one sample for sets too small to resolve that), the success
RATE per flavor and pooled within 0.5 %, and -- full mode, >= 2000 pooled samples -- the 95 % interval of the paired
difference of the two rates inside +-0.5 % as well; every number is printed.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import test_success_bits as tsb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (fixture file, flavor).  The first two are required; further sets (``asr_base_<flavor>_<tag>.json``: independent draws
# with another seed -- other images, questions, schedules) are picked up when committed.
import glob  # noqa: E402
FIXTURES = [("asr_base_vlmo.json", "vlmo"), ("asr_base_albef.json", "albef")]
TIE = 1e-3          # oracle decision margin (gap between the victim's two leading answers) below which a sample is a tie
RESULTS = {}        # fixture -> dict(flavor, n, want bits, got bits, margins, sample ids), for the pooled test
FULL = os.environ.get("VQA_ASR_FULL", "") not in ("", "0")
# driver-run suite: the first K samples of every fixture file (round 6: 1 775 of the pool's 3 366 samples, ~ 4.5 min of
# the suite's 15-minute limit; rounds 4-5: 760); the whole pool runs with VQA_ASR_FULL=1 (profiles/r05/asr_full_report.json)
SUBSET = dict(vlmo=150, albef=96)
SUBSET_SMALL = dict(vlmo=32, albef=16)   # ... of the many small draws (< 100 samples, tools/asr_box_round.sh)
_MODELS = {}        # (flavor, answer-set size of the ALBEF victim) -> (white on the GPU, black on the GPU, adapters, cfg)


def _models(flavor, k, dev):
    """The frozen pair of a flavor, built ONCE for all fixture files (same seeds in every file; the ALBEF victim's
    answer list depends on its size): most of a small set's wall time is the CPU-side construction of the base models."""
    key = (flavor, k if flavor == "albef" else None)
    if key not in _MODELS:
        cfg_kw = dict(n_answers=k, k_test=min(128, k)) if flavor == "albef" else {}
        white, black, adapters_cls, _, cfg = tsb.build(flavor, "base", **cfg_kw)
        _MODELS[key] = (white.to(dev), black.to(dev), adapters_cls, cfg)
    return _MODELS[key]
FIXTURES += sorted((os.path.basename(p), os.path.basename(p).split("_")[2])
                   for p in glob.glob(os.path.join(ROOT, "tests", "golden", "asr_base_*_*.json")))


def _fixture(name, flavor):
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(path):
        pytest.fail("missing fixture {} (python tests/golden/make_asr_fixture.py --flavor {})".format(path, flavor))
    rec = json.load(open(path))
    assert rec["flavor"] == flavor
    return rec


@pytest.mark.parametrize("name,flavor", FIXTURES, ids=[f[0][9:-5] for f in FIXTURES])
def test_base_size_success_bits_match_the_recorded_oracle(name, flavor):
    from vqattack_amd.attack import text_update
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    rec = _fixture(name, flavor)
    # informative: both outcomes occur often.  The generator picks the answer-set size whose oracle ASR lies inside
    # 0.3 .. 0.7; a draw of a few dozen samples may have no such size (the rate moves in steps of 1 / n with n and jumps
    # between neighbouring answer-set sizes): then the size closest to 0.5, which must still lie inside 0.2 .. 0.8
    assert rec["size"] == "base" and 0.2 <= rec["oracle_asr"] <= 0.8, "the fixture must be informative"
    n_all, k = rec["n"], rec["n_answers"]               # n_all: size of the seeded draw that regenerates the inputs
    n_have = rec.get("n_scored", n_all)                 # ... of which the oracle attacked and scored a prefix
    n = n_have if FULL else min(n_have, (SUBSET if n_have >= 100 else SUBSET_SMALL)[flavor])
    dev = torch.device("cuda", 0)
    assert (rec["white_seed"], rec["black_seed"]) == (3, 4)          # tsb.build's seeds: one model pair per flavor
    white_gpu, black_gpu, adapters_cls, cfg = _models(flavor, k, dev)
    shape = rec["shape"]
    ids, masks, att, tasks, _, images, eta = tsb.make_samples(flavor, cfg, n=n_all, seed=rec["seed"],
                                                             words=tuple(shape["words"]), max_att=shape["max_att"],
                                                             text_len=shape["text_len"])
    proposals = [[(int(p), [int(v) for v in vs]) for p, vs in row] for row in rec["proposals"]]
    sim = text_update.BagOfEmbeddingsSimilarity(seed=rec["sim_seed"])
    attack = BatchedVQAttack(adapters_cls(white_gpu), flavor, white_gpu.embedding_tables(),
                             AttackConfig(budget=rec["budget"], sanity_checks=True, sim_threshold=rec["sim_threshold"]),
                             similarity_fn=sim)
    answer = (lambda *a: black_gpu.vqa_answer(*a, n_answers=k)) if flavor == "vlmo" else black_gpu.vqa_answer
    batch = 50 if flavor == "vlmo" else 32
    got_clean, got_after, got_ids = [], [], []
    for lo in range(0, n, batch):
        sl = slice(lo, min(lo + batch, n))
        img, tid, tm = images[sl].to(dev), ids[sl].to(dev), masks[sl].to(dev)
        got_clean += answer(img, tid, tm).cpu().tolist()
        res = attack.attack_mixed(img, tid, tm, att[sl].to(dev), init_eta=eta[sl].to(dev), proposals=proposals[sl],
                                  tasks=tasks[sl])
        got_after += answer(res.adv_images, res.adv_text_ids, tm).cpu().tolist()
        got_ids.append(res.adv_text_ids.cpu())
        del res, img
    got_bits = [int(a != c) for a, c in zip(got_after, got_clean)]
    want_bits, margins = rec["success_bits"], rec["adversarial_margins"]
    differ = [s for s in range(n) if got_bits[s] != want_bits[s]]
    ans_differ = [s for s in range(n) if got_after[s] != rec["adversarial_answers"][s]]
    id_rows = int((torch.cat(got_ids) != torch.tensor(rec["adv_text_ids"][:n])).any(dim=1).sum())
    print("{} base: n = {}{}, closed answer set of {}, oracle ASR {:.4f}, product ASR {:.4f}; {} success bits differ "
          "({:.2%}), {} adversarial answer indices differ, {} rows of substituted ids differ; oracle margins of the "
          "differing samples: {}".format(flavor, n, "" if n == n_have else " (first {} of {})".format(n, n_have), k,
                                         float(np.mean(want_bits[:n])), float(np.mean(got_bits)), len(differ),
                                         len(differ) / n, len(ans_differ), id_rows,
                                         [round(margins[s], 5) for s in sorted(set(differ + ans_differ))]))
    assert got_clean == rec["clean_answers"][:n], "the victim's clean answers differ"
    assert id_rows == 0, "substituted token ids differ in {} samples".format(id_rows)
    # A bit may differ only where the victim is TIED between two answers (the oracle's own decision margin is below TIE:
    # the two 40-step sign trajectories agree in ~99.5 % of the pixels, which moves the victim's scores by a few 1e-4 --
    # DESIGN.md section 6 shows one such sample logit by logit), and in no more than 0.5 % of the samples; a set smaller
    # than 200 samples cannot resolve 0.5 % (one sample is more), so one tied sample is the floor.
    not_tied = [s for s in differ if abs(margins[s]) >= TIE]
    assert not not_tied, "success bits differ on samples the victim decides clearly: {} (margins {})".format(
        not_tied, [margins[s] for s in not_tied])
    assert len(differ) <= max(1, int(0.005 * n)), "{} of {} success bits differ: samples {}".format(len(differ), n, differ)
    assert 0 < sum(got_bits) < n
    RESULTS[name] = dict(flavor=flavor, n=n, n_in_file=n_have, seed=rec["seed"], n_answers=k, want=want_bits[:n],
                         got=got_bits, margins=margins[:n], differ=differ)
    if os.environ.get("VQA_ASR_SETS_LOG"):           # one line per set as it finishes (tools/asr_box_round.sh reads it)
        with open(os.environ["VQA_ASR_SETS_LOG"], "a") as f:
            f.write(json.dumps(dict(fixture=name, flavor=flavor, n=n, oracle_asr=sum(want_bits[:n]) / n,
                                    product_asr=sum(got_bits) / n, differ=differ,
                                    margins=[margins[s] for s in differ])) + "\n")


def paired_difference(want, got):
    """(difference of the two success rates, half-width of its 95 % interval) for PAIRED bits: the samples are the same
    on both sides, so only the discordant pairs carry variance (n10: product succeeds where the oracle fails, n01 the
    other way): d = (n10 - n01) / n, var(d) = (n10 + n01 - (n10 - n01)^2 / n) / n^2."""
    n = len(want)
    n10 = sum(1 for w, g in zip(want, got) if g and not w)
    n01 = sum(1 for w, g in zip(want, got) if w and not g)
    d = (n10 - n01) / n
    var = max(n10 + n01 - (n10 - n01) ** 2 / n, 0.0) / n ** 2
    return d, 1.96 * var ** 0.5, n10, n01


def test_pooled_success_rate_within_half_a_percent():
    """north_star: attack-success rate within +-0.5 % of the reference's -- per flavor and over every base-size sample of
    the sets above."""
    missing = [f[0] for f in FIXTURES if f[0] not in RESULTS]
    if missing:
        pytest.skip("needs the per-set tests of this module to have run first (missing: {})".format(missing))
    report = dict(mode="full" if FULL else "subset (first {} VLMO / {} ALBEF samples of every file, {} / {} of a draw of "
                  "fewer than 100)".format(SUBSET["vlmo"], SUBSET["albef"], SUBSET_SMALL["vlmo"], SUBSET_SMALL["albef"]),
                  tie_margin=TIE, sets={}, pooled={})
    for name, r in RESULTS.items():
        report["sets"][name] = dict(flavor=r["flavor"], n=r["n"], n_in_file=r["n_in_file"], seed=r["seed"],
                                    n_answers=r["n_answers"], oracle_asr=sum(r["want"]) / r["n"],
                                    product_asr=sum(r["got"]) / r["n"],
                                    differing_samples=[dict(sample=s, oracle_bit=r["want"][s], product_bit=r["got"][s],
                                                            oracle_margin=r["margins"][s]) for s in r["differ"]])
    for flavor in ("vlmo", "albef", None):
        rows = [r for r in RESULTS.values() if flavor in (None, r["flavor"])]
        want = [b for r in rows for b in r["want"]]
        got = [b for r in rows for b in r["got"]]
        n, diff = len(want), sum(len(r["differ"]) for r in rows)
        d, half, n10, n01 = paired_difference(want, got)
        print("{}: {} samples in {} sets, oracle ASR {:.4f}, product ASR {:.4f}, difference {:+.4%} +- {:.4%} (95 %, paired: "
              "{} bits 0 -> 1, {} bits 1 -> 0), {} bits differ ({:.2%})".format(
                  flavor or "all", n, len(rows), sum(want) / n, sum(got) / n, d, half, n10, n01, diff, diff / n))
        report["pooled"][flavor or "all"] = dict(n=n, sets=len(rows), oracle_asr=sum(want) / n, product_asr=sum(got) / n,
                                                 difference=d, ci95_half_width=half, product_only=n10, oracle_only=n01,
                                                 differing_bits=diff, differing_rate=diff / n)
        # per flavor and pooled (a flavor with fewer than 200 samples could not resolve 0.5 %: one sample would be more)
        assert n < 200 or (abs(d) <= 0.005 and diff / n <= 0.005), flavor
        if FULL and n >= 2000:      # the interval, not just the point estimate, inside the north star's +-0.5 %
            assert abs(d) + half <= 0.005, "{}: difference {:+.4%} +- {:.4%}".format(flavor or "all", d, half)
    _MODELS.clear()
    torch.cuda.empty_cache()
    out = os.environ.get("VQA_ASR_REPORT")
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        json.dump(report, open(out, "w"), indent=1)
