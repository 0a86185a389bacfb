"""Same module path as the reference's cleverhans/torch/utils.py; MI355X implementation in vqattack_amd.utils."""
from vqattack_amd.utils import (clip_eta, get_or_guess_labels, optimize_linear,  # noqa: F401
                                zero_out_clipped_grads)
