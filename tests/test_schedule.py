"""Block schedule (a11): the shipped scheduler against the oracle restatement and hand-evaluated values of
``ALBEF_attack/adv_attack.py:229-239`` (count = words + 1; q = int(40/count); even -> [q]*count else [q-1]*count;
last += 40 - sum)."""
import pytest

from oracle.text_scoring import iter_schedule as oracle_schedule
from vqattack_amd.attack.schedule import bucket_by_schedule, gradient_steps, iter_schedule

HAND = {0: [], 1: [20, 20], 2: [12, 12, 16], 3: [10, 10, 10, 10], 4: [8, 8, 8, 8, 8], 5: [6, 6, 6, 6, 6, 10],
        7: [4] * 7 + [12], 9: [4] * 9 + [4], 12: [2] * 12 + [16], 19: [2] * 19 + [2], 20: [0] * 20 + [40],
        39: [0] * 39 + [40], 40: [0] * 40 + [40]}


@pytest.mark.parametrize("words", range(0, 42))
def test_schedule_matches_oracle(words):
    got = iter_schedule(words)
    assert got == oracle_schedule(words)
    if words:
        assert sum(got) == 40 and len(got) == words + 1
        assert all(b % 2 == 0 for b in got[:-1])
    assert gradient_steps(words) == 40 + words


@pytest.mark.parametrize("words,blocks", sorted(HAND.items()))
def test_schedule_hand_values(words, blocks):
    assert iter_schedule(words) == blocks


def test_bucketing():
    assert bucket_by_schedule([3, 5, 3, 0, 5, 5]) == {0: [3], 3: [0, 2], 5: [1, 4, 5]}
