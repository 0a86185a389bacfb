"""The C-ABI library loads on a machine without a GPU and exports every symbol include/vqattack_hip.h declares."""
import ctypes
import os
import re

from vqattack_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vqattack_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef VQA_TUNING.*?#endif", "", text, flags=re.S)      # tools-only build, not the shipped ABI
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
    assert lib.vqa_abi_version() == 4
    assert lib.vqa_neg_cos_partials() > 0
    assert lib.vqa_reduce_ws_bytes(4, 3 * 384 * 384) > 0
    assert lib.vqa_error_string(-1).decode().startswith("a required pointer")
    assert not hasattr(lib, "vqa_set_option"), "the shipped library has one fixed variant per kernel: no knobs"


def test_argument_validation_returns_error_codes_without_touching_a_gpu():
    """Every entry point validates its arguments before the first HIP call, so the error paths are testable here."""
    import ctypes as C
    lib = _hip.lib()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    off = C.c_void_p(p.value + 2)                      # not 4-byte aligned
    null = None
    ERR_NULL, ERR_SHAPE, ERR_ALIGN = -1, -2, -3
    assert lib.vqa_linf_step(null, p, p, p, 16, 0.01, 0.1, -1.0, 1.0, 1, null, null) == ERR_NULL
    assert lib.vqa_linf_step(p, p, p, p, 16, 0.01, 0.1, -1.0, 1.0, 3, null, null) == ERR_NULL      # flag required
    assert lib.vqa_linf_step(off, p, p, p, 16, 0.01, 0.1, -1.0, 1.0, 1, null, null) == ERR_ALIGN
    assert lib.vqa_linf_step(p, p, p, p, 0, 0.01, 0.1, -1.0, 1.0, 1, null, null) == 0            # empty: no launch
    assert lib.vqa_linf_fgm(p, null, p, 16, 0.01, -1.0, 1.0, 1, null, null) == ERR_NULL
    assert lib.vqa_linf_init(p, null, null, 16, 0.1, -1.0, 1.0, 1, null, null) == ERR_NULL
    assert lib.vqa_sumsq_per_sample(p, null, p, 70000, 4, p, null) == ERR_SHAPE                  # batch > 65535
    assert lib.vqa_sumsq_per_sample(p, null, p, 2, 4, null, null) == ERR_NULL                    # workspace missing
    assert lib.vqa_scale_per_sample(p, p, null, p, 2, 8, 0.5, 9, null, null) == ERR_SHAPE        # unknown kind
    assert lib.vqa_scale_per_sample(p, p, null, p, 2, 8, 0.5, 2, null, null) == ERR_NULL         # L1 needs ties
    assert lib.vqa_neg_cos_rows(p, p, null, p, null, 1, 2, 2, 6, 12, 6, 12, 6, 0, 0, 1.0, 1e-6, null, 0, null) == ERR_SHAPE   # D % 4
    assert lib.vqa_neg_cos_rows(p, p, null, p, null, 1, 2, 2, 4096, 0, 0, 0, 0, 0, 0, 1.0, 1e-6, null, 0, null) == ERR_SHAPE  # D > 2048
    assert lib.vqa_neg_cos_rows(p, p, null, p, null, 1, 2, 2, 8, 18, 8, 16, 8, 0, 0, 1.0, 1e-6, null, 0, null) == ERR_SHAPE   # stride % 4
    assert lib.vqa_neg_cos_rows(p, p, null, null, null, 1, 2, 2, 8, 16, 8, 16, 8, 0, 0, 1.0, 1e-6, null, 0, null) == ERR_NULL
    assert lib.vqa_ce_rows(p, 8, p, 9, 2, 8, -100, 0, p, null, p, 1.0, null, 0, null, null, null) == ERR_SHAPE     # K > 8
    assert lib.vqa_ce_rows(p, 4, p, 1, 2, 8, -100, 0, p, null, p, 1.0, null, 0, null, null, null) == ERR_SHAPE     # row stride < V
    assert lib.vqa_ce_rows(p, 8, p, 1, 2, 8, -100, 0, p, null, p, 1.0, null, 0, null, p, null) == ERR_NULL         # row_state needs grad
    assert lib.vqa_gather_rows(p, null, p, 1, 4, 2, 8, null) == ERR_NULL
    assert lib.vqa_cand_dir_sim(p, p, p, p, p, 1e-12, p, p, p, p, 1, 4, 2, 6, null) == ERR_SHAPE
    assert lib.vqa_embed_tokens(p, p, p, p, p, 1e-12, null, 1, p, 8, null) == ERR_NULL
    assert lib.vqa_resize_bicubic_h_u8(p, 4, 4, 7, p, p, 5, 8, p, null) == ERR_SHAPE             # channels > 4
    assert lib.vqa_resize_bicubic_v_normalize(p, 4, 4, 3, null, null, 0, 8, 0.5, 0.5, p, null) == ERR_NULL   # needs taps
    assert lib.vqa_resize_bicubic_v_normalize(p, 4, 4, 3, p, p, 5, 8, 0.5, 0.0, p, null) == ERR_SHAPE        # std == 0
    assert lib.vqa_ce_scratch_floats(8, 64) >= 8 * 64 + 1 and lib.vqa_ce_scratch_floats(0, 1) == 0
    for code in (ERR_NULL, ERR_SHAPE, ERR_ALIGN):
        assert lib.vqa_error_string(code)
