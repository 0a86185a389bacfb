"""Multi-head attention core of the frozen white boxes: one place that decides which kernel runs.

Tensors are (B, S, H, d) -- the layout a ``Linear`` + ``reshape`` produces, so neither side of the attention needs a
transpose copy.  On the GPU, for the head size of every BASELINE configuration (d = 64: VLMo-base / -large, ALBEF), the
exact-fp32 MFMA kernels of ``csrc/attn.hip`` run (``vqattack_amd.attention``) and a missing HIP library is an error,
not a fallback.  Other head sizes (the test-sized models) and host tensors (the white boxes also run on the CPU as the
frozen model behind the oracle in ``tests/``) go through PyTorch's ``scaled_dot_product_attention``.
"""
import torch
import torch.nn.functional as F

from .. import attention as _attn


def uses_hip(t):
    return t.is_cuda and t.dtype == torch.float32 and t.shape[-1] == _attn.HEAD_DIM


def mha(q, k, v, bias=None):
    """softmax(q k^T / sqrt(d) + bias) v; q (B, Sq, H, d), k / v (B, Sk, H, d), bias broadcastable to
    (B, H, Sq, Sk) -> (B, Sq, H, d)."""
    if uses_hip(q):
        return _attn.attention(q, k, v, bias)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias)
    return o.transpose(1, 2)


def mha_packed(qkv, bias=None):
    """Self-attention on a packed projection output qkv (B, S, 3, H, d) -> (B, S, H, d)."""
    if uses_hip(qkv):
        return _attn.self_attention_packed(qkv, bias)
    return mha(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias)
