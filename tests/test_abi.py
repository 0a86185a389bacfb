"""The C-ABI library loads on a machine without a GPU and exports every symbol include/vqattack_hip.h declares."""
import ctypes
import os
import re

from vqattack_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vqattack_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(vqa_[a-z0-9_]+)\s*\(", text))


def test_header_and_binding_agree():
    assert _declared() == set(_hip.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.LIB_PATH), "run `python -m vqattack_amd.build` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_compute_entry_points():
    """Only the argument-free queries are callable without a GPU."""
    lib = _hip.lib()
    assert lib.vqa_abi_version() == 1
    assert lib.vqa_neg_cos_partials() > 0
    assert lib.vqa_reduce_ws_bytes(4, 3 * 384 * 384) > 0
    assert lib.vqa_error_string(-1).decode().startswith("a required pointer")
    assert lib.vqa_set_option(99, 0) == -2
