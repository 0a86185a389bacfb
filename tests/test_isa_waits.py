"""The compiled hot loops keep their memory-instruction structure (no GPU needed: hipcc cross-compiles here).

gfx950 counts a wave's outstanding loads and stores in ONE in-order counter; behind a branch around a load or store the
compiler's `s_waitcnt vmcnt(N)` degrades to "wait for everything" ("One counter for loads and
stores").  The kernels were restructured in round 6 so that their steady-state loops have exact waits; a source edit or
a compiler update that brings a conservative wait back changes no result and would go unnoticed -- these tests read
the loop's load / store / wait sequence from the assembly (`tools/isa_loop_mix.py`).
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="needs hipcc")


@pytest.fixture(scope="module")
def loops():
    import isa_loop_mix
    found = {}
    for src in ("linf.hip", "loss.hip", "attn.hip"):
        for name, meta, body in isa_loop_mix.kernels(isa_loop_mix.assembly(src)):
            found[name] = (meta, isa_loop_mix.memory_sequence(body))
    return found


def _runs(seq, symbol):
    """lengths of the maximal runs of `symbol` in the sequence"""
    out, n = [], 0
    for s in seq + [None]:
        if s == symbol:
            n += 1
        elif n:
            out.append(n)
            n = 0
    return out


def test_step_kernel_issues_a_whole_tile_before_its_first_wait(loops):
    meta, seq = loops["vqa::stream4_kernel<vqa::StepOp, 4, 5>"]
    assert seq[:12] == ["L"] * 12, seq                    # 3 streams x 4 tiles, back to back
    assert seq[12].startswith("w") and seq[12] != "w0", seq
    assert _runs(seq, "S") == [4], seq                    # the four stores leave together
    assert meta["ScratchSize"] == "0"


def test_loss_kernel_for_whole_chunk_rows_keeps_two_rows_in_flight(loops):
    for name in ("vqa::neg_cos_rows_full_kernel<3, true, 7>", "vqa::neg_cos_rows_full_kernel<4, true, 7>"):
        meta, seq = loops[name]                # (all loops of the kernel: the row loop and the fold's small ones)
        nch = int(name.split("<")[1].split(",")[0])
        text = " ".join(seq)
        assert "S w0 S" not in text and "S w1 S" not in text, (name, seq)   # no store acknowledged before the next piece
        assert _runs(seq, "S").count(nch) >= 2, (name, seq)                 # both rows of the ping-pong: stores together
        prefetches = [i for i in range(len(seq)) if seq[i:i + 2 * nch] == ["L"] * (2 * nch)]
        assert prefetches, (name, seq)
        for i in prefetches:                                                # what follows a row's loads is not "wait for all"
            nxt = next((x for x in seq[i + 2 * nch:] if x.startswith("w")), None)
            assert nxt is not None and nxt != "w0", (name, seq)
        assert meta["ScratchSize"] == "0"
    # the general kernel, for contrast, still shows the pattern (if this ever fails the note in loss.hip is out of date)
    _, general = loops["vqa::neg_cos_rows_kernel<3, true, true, 4>"]
    assert "S w0 S" in " ".join(general)


def test_attention_stores_follow_the_lds_publish_and_nothing_waits_for_them_inside_the_tile(loops):
    for name in ("vqa::attn_fwd_kernel<true, true, false>", "vqa::attn_bwd_dkv_kernel<true, true, true, false>"):
        meta, seq = loops[name]
        body = [s for s in seq if s != "|"]
        last_store = max(i for i, s in enumerate(body) if s == "S")
        assert not any(s.startswith("w") for s in body[body.index("S"):last_store]), (name, seq)   # stores in one group ...
        assert all(s == "S" for s in body[body.index("S"):]), (name, seq)                          # ... at the tile's end
        # the tile prefetch at the top of the loop is issued without a wait between its loads (a register allocation that
        # puts the prefetch addresses into the previous tile's load registers brings write-after-write waits -- for the
        # previous tile's stores -- in between: seen once with non-temporal stores, profiles/r06/attention/ab_8...)
        first = body.index("L")
        eighth = [i for i, x in enumerate(body) if x == "L"][7]
        assert not any(x.startswith("w") for x in body[first:eighth]), (name, seq)
        assert meta["ScratchSize"] == "0"
    # occupancy the launch shapes rely on: 3 waves per SIMD forward, 2 for the key-block kernel
    assert loops["vqa::attn_fwd_kernel<true, true, false>"][0]["Occupancy"] == "3"
    assert loops["vqa::attn_bwd_dkv_kernel<true, true, true, false>"][0]["Occupancy"] == "2"
