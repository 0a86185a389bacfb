"""Build-container integrity check: the committed fixtures ARE what the reference produces.

Runs only where ``/root/reference`` exists (the build container; never on the GPU box): both generators are re-run into
a scratch directory -- importing the reference's cleverhans modules and executing its ast-compiled orchestrator methods
-- and every array / record must equal the committed ``tests/golden/*`` files.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")


def _run(code, tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT)
    proc = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]


def _same_npz(a, b):
    za, zb = np.load(a), np.load(b)
    assert set(za.files) == set(zb.files)
    for k in za.files:
        assert np.array_equal(za[k], zb[k], equal_nan=True), k


def test_cleverhans_golden_reproduces_from_the_reference(tmp_path):
    out = str(tmp_path / "cleverhans_golden.npz")
    _run("from tests.golden import make_golden as m; m.main({!r})".format(out), tmp_path)
    _same_npz(out, os.path.join(GOLD, "cleverhans_golden.npz"))


def test_text_golden_reproduces_from_the_reference(tmp_path):
    npz, js = str(tmp_path / "text_golden.npz"), str(tmp_path / "text_golden.json")
    _run("from tests.golden import make_text_golden as m; m.main({!r}, {!r})".format(npz, js), tmp_path)
    _same_npz(npz, os.path.join(GOLD, "text_golden.npz"))
    with open(js) as fa, open(os.path.join(GOLD, "text_golden.json")) as fb:
        assert json.load(fa) == json.load(fb)


def test_encoder_golden_reproduces_from_the_reference(tmp_path):
    """The white-box encoder fixtures: the reference's Mlp / Attention / Block / MultiWayTransformer / VisionTransformer /
    BertEncoder classes and the VLMo attack closures, executed from source again, give the committed arrays."""
    npz, js = str(tmp_path / "encoder_golden.npz"), str(tmp_path / "encoder_golden.json")
    _run("from tests.golden import make_encoder_golden as m; m.main({!r}, {!r})".format(npz, js), tmp_path)
    _same_npz(npz, os.path.join(GOLD, "encoder_golden.npz"))
    with open(js) as fa, open(os.path.join(GOLD, "encoder_golden.json")) as fb:
        assert json.load(fa) == json.load(fb)
