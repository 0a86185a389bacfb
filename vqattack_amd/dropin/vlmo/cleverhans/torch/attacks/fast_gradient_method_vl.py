"""Same module path as the reference's attacks/fast_gradient_method_vl.py (vlmo copy)."""
import functools

from vqattack_amd import attacks as _impl

fast_gradient_method = functools.wraps(_impl.fast_gradient_method_vl)(
    functools.partial(_impl.fast_gradient_method_vl, flavor="vlmo"))
