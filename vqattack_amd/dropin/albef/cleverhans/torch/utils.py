"""Same module path as the reference's cleverhans/torch/utils.py; MI355X implementation in vqattack_amd.utils."""
from vqattack_amd.utils import clip_eta, optimize_linear  # noqa: F401
