"""Same module path as the reference's attacks/projected_gradient_descent.py (albef copy)."""
import functools

from vqattack_amd import attacks as _impl

projected_gradient_descent = functools.wraps(_impl.projected_gradient_descent)(
    functools.partial(_impl.projected_gradient_descent, flavor="albef"))
