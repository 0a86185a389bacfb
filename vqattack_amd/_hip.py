"""ctypes binding of ``lib/libvqattack_hip.so`` (C ABI declared in ``include/vqattack_hip.h``).

PyTorch is used for device memory and streams only: every wrapper takes ``torch.Tensor`` arguments, checks
that they live on a HIP device in the layout the kernel assumes, and launches on torch's CURRENT stream of
that device.  There is no CPU path: a missing library, a CPU tensor or a failed launch raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvqattack_hip.so")
# tools/ only: the -DVQA_TUNING build (launch-shape knobs + A/B kernel variants), selected with VQA_TUNING_LIB=1
TUNING_LIB_PATH = os.path.join(_HERE, "lib", "libvqattack_hip_tuning.so")

VQA_CLIP = 1
VQA_CHECK_RANGE = 2
VQA_FLAG_RANGE = 1        # bits of a flag word
VQA_FLAG_BAD_LABEL = 2
VQA_FLAG_DEGENERATE = 4   # optimize_linear's self-check would fail (all-zero / NaN L1 gradient, non-finite L2 norm)
ABI_VERSION = 4

_c_float_p = ctypes.c_void_p
_sz = ctypes.c_size_t
_f = ctypes.c_float
_i = ctypes.c_int
_u = ctypes.c_uint
_l = ctypes.c_long
_p = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/vqattack_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "vqa_abi_version": (_i, []),
    "vqa_error_string": (ctypes.c_char_p, [_i]),
    "vqa_linf_init": (_i, [_p, _p, _p, _sz, _f, _f, _f, _u, _p, _p]),
    "vqa_linf_fgm": (_i, [_p, _p, _p, _sz, _f, _f, _f, _u, _p, _p]),
    "vqa_linf_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _u, _p, _p]),
    "vqa_linf_project": (_i, [_p, _p, _p, _sz, _f, _f, _f, _u, _p]),
    "vqa_clip_eta_linf": (_i, [_p, _p, _sz, _f, _p]),
    "vqa_optimize_linear_linf": (_i, [_p, _p, _sz, _f, _p]),
    "vqa_zero_out_clipped_grads": (_i, [_p, _p, _p, _sz, _f, _f, _p]),
    "vqa_reduce_ws_bytes": (_sz, [_i, _sz]),
    "vqa_sumsq_per_sample": (_i, [_p, _p, _p, _i, _sz, _p, _p]),
    "vqa_absmax_ties_per_sample": (_i, [_p, _p, _p, _i, _sz, _p, _p]),
    "vqa_l2_fgm": (_i, [_p, _p, _p, _p, _i, _sz, _f, _f, _f, _u, _p, _p]),
    "vqa_l2_project": (_i, [_p, _p, _p, _p, _i, _sz, _f, _f, _f, _u, _p]),
    "vqa_l1_fgm": (_i, [_p, _p, _p, _p, _p, _i, _sz, _f, _f, _f, _u, _p, _p]),
    "vqa_scale_per_sample": (_i, [_p, _p, _p, _p, _i, _sz, _f, _i, _p, _p]),
    "vqa_neg_cos_partials": (_i, []),
    "vqa_neg_cos_rows": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _i, _l, _l, _l, _l, _l, _l, _f, _f, _p, _i, _p]),
    "vqa_neg_cos_max_layers": (_i, []),
    "vqa_neg_cos_rows_multi": (_i, [_p, _p, _p, _i, _p, _p, _l, _l, _l, _i, _l, _l, _l, _l, _l, _l, _f, _f, _p, _i, _p]),
    "vqa_sum_partials": (_i, [_p, _i, _p, _i, _f, _p]),
    "vqa_ce_max_label_sets": (_i, []),
    "vqa_ce_scratch_floats": (_l, [_i, _l]),
    "vqa_ce_rows": (_i, [_p, _l, _p, _i, _l, _i, _l, _l, _p, _p, _p, _f, _p, _i, _p, _p, _p]),
    "vqa_gather_rows": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vqa_cand_dir_sim": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vqa_embed_tokens": (_i, [_p, _p, _p, _p, _p, _f, _p, _i, _p, _i, _p]),
    "vqa_greedy_accept": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "vqa_attn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _p, _i, _p, _p]),
    "vqa_attn_split_ws_floats": (_l, [_i, _i, _i, _i, _i]),
    "vqa_attn_scores_floats": (_l, [_i, _i, _i, _i]),
    "vqa_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p, _p]),
    "vqa_attn_bwd_ws_floats": (_l, [_i, _i, _i, _i]),
    "vqa_ln_fwd": (_i, [_p] * 13 + [_l, _i, _l, _l, _f, _p]),
    "vqa_ln_bwd": (_i, [_p] * 13 + [_l, _i, _l, _l, _p]),
    "vqa_gelu_fwd": (_i, [_p, _p, _sz, _p]),
    "vqa_gelu_bwd": (_i, [_p, _p, _p, _sz, _p]),
    "vqa_resize_bicubic_h_u8": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "vqa_resize_bicubic_v_normalize": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _f, _f, _p, _p]),
}

_lib = None
_lock = threading.Lock()


class HipExtensionError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the kernel library and type its entry points.  Raises if it is absent (no fallback)."""
    tuning = os.environ.get("VQA_TUNING_LIB", "") not in ("", "0")
    if path is None:
        path = TUNING_LIB_PATH if tuning else LIB_PATH
    if not os.path.exists(path):
        raise HipExtensionError(
            "HIP kernel library not found at {} -- build it with `python -m vqattack_amd.build` "
            "(or __graft_entry__.build()); vqattack_amd has no CPU or eager fallback".format(path))
    if tuning and path == TUNING_LIB_PATH:
        from . import build as _build          # an A/B run on kernels older than the sources would measure nothing
        if _build._stale(path):
            raise HipExtensionError("{} is older than vqattack_amd/csrc: rebuild it with `python -m vqattack_amd.build "
                                    "--tuning`".format(path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.vqa_abi_version() != ABI_VERSION:
        raise HipExtensionError("{} implements C ABI version {}, this package binds version {}: rebuild it with "
                                "`python -m vqattack_amd.build --force`".format(path, lib.vqa_abi_version(), ABI_VERSION))
    # tools/ only: VQA_OPTIONS="1=3,0=8" -> vqa_set_option(1, 3); vqa_set_option(0, 8).  The shipped library has no
    # knobs (one launch shape per kernel, fixed at compile time): the overrides need the tuning build
    options = [item for item in os.environ.get("VQA_OPTIONS", "").split(",") if item]
    if options:
        if not hasattr(lib, "vqa_set_option"):
            raise HipExtensionError("VQA_OPTIONS needs the tuning build: `python -m vqattack_amd.build --tuning` and "
                                    "VQA_TUNING_LIB=1 ({} has no vqa_set_option)".format(path))
        lib.vqa_set_option.restype, lib.vqa_set_option.argtypes = _i, [_i, _i]
        for item in options:
            opt, val = item.split("=")
            if lib.vqa_set_option(int(opt), int(val)) != 0:
                raise HipExtensionError("bad VQA_OPTIONS entry '{}'".format(item))
    return lib


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load_library()
    return _lib


def set_option(option, value):
    """tools/ only: a launch-shape knob of the TUNING build (VQA_TUNING_LIB=1).  Returns False when the loaded library is
    the shipped one, which has no knobs (one variant per kernel, fixed at compile time); raises on a bad value."""
    handle = lib()
    if not hasattr(handle, "vqa_set_option"):
        return False
    handle.vqa_set_option.restype, handle.vqa_set_option.argtypes = _i, [_i, _i]
    if handle.vqa_set_option(int(option), int(value)) != 0:
        raise HipExtensionError("vqa_set_option({}, {}) rejected".format(option, value))
    return True


def check(code, what):
    if code != 0:
        msg = lib().vqa_error_string(code)
        raise HipExtensionError("{} failed: {} ({})".format(what, msg.decode() if msg else "?", code))


def stream_for(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def dev_f32(t, name, contiguous=True):
    """Validate a tensor the kernels will read or write through its raw pointer."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("{} must be a torch.Tensor, got {}".format(name, type(t)))
    if not t.is_cuda:
        raise HipExtensionError(
            "{} lives on '{}': the VQAttack hot path of vqattack_amd runs on an MI355X (HIP) device only; "
            "there is no CPU fallback".format(name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("{} must be float32, got {}".format(name, t.dtype))
    if contiguous and not t.is_contiguous():
        raise ValueError("{} must be contiguous".format(name))
    return t


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def same_device(*ts):
    devs = {t.device for t in ts if t is not None}
    if len(devs) > 1:
        raise ValueError("tensors are on different devices: {}".format(sorted(map(str, devs))))
