"""Iteration schedule of the joint attack (row a11 of SURVEY.md section 8a).

Reference: ``cal_text_attack_list``, ``ALBEF_attack/adv_attack.py:229-239`` (identical in
``vlmo/modules/vlmo_module.py:1545-1556``): a question with ``w`` substitutable words is attacked in ``w + 1`` PGD
blocks whose lengths sum to the 40-step image budget; between consecutive blocks one image+text probe step
(``projected_gradient_descent_vl`` with ``nb_iter=1``) drives the word substitutions, so a sample takes
``40 + w`` white-box gradient steps in total.
"""
IMAGE_STEP_BUDGET = 40


def iter_schedule(n_words, budget=IMAGE_STEP_BUDGET):
    """PGD steps per block; ``[]`` when the question has no substitutable word (single 40-step PGD instead)."""
    if n_words <= 0:
        return []
    count = n_words + 1
    per = int(budget / count)
    if per % 2:
        per -= 1                       # the reference keeps block lengths even (dual-loss blocks run per/2 iterations)
    blocks = [per] * count
    blocks[-1] += budget - sum(blocks)
    return blocks


def gradient_steps(n_words, budget=IMAGE_STEP_BUDGET):
    """White-box forward/backward passes one sample costs (image steps + probe steps)."""
    return budget + max(n_words, 0)


def bucket_by_schedule(n_words_per_sample):
    """Group sample indices so that every bucket shares one schedule (samples of a bucket run as one batch)."""
    buckets = {}
    for idx, w in enumerate(n_words_per_sample):
        buckets.setdefault(int(w), []).append(idx)
    return dict(sorted(buckets.items()))
