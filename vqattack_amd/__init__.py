"""vqattack_amd -- MI355X-native implementation of VQAttack's PGD image(+text) perturbation hot path.

Host side in Python (mirrors the reference's cleverhans-style operator API), device side in hand-written HIP for
gfx950 behind a C ABI (``include/vqattack_hip.h`` -> ``vqattack_amd/lib/libvqattack_hip.so``).
"""
from . import attacks, ops, utils  # noqa: F401
from .attacks import (ALBEF, VLMO, fast_gradient_method, fast_gradient_method_vl,  # noqa: F401
                      projected_gradient_descent, projected_gradient_descent_vl)
from .utils import clip_eta, optimize_linear  # noqa: F401

__version__ = "0.1.0"
