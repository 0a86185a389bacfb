"""Multi-head attention core of the frozen white boxes: one place that decides which kernel runs.

Tensors are (B, S, H, d) -- the layout a ``Linear`` + ``reshape`` produces, so neither side of the attention needs a
transpose copy.  On the GPU every fp32 attention of the white boxes runs on the exact-fp32 MFMA kernels of
``csrc/attn.hip`` (``vqattack_amd.attention``); a missing HIP library is an error, not a fallback.  The kernels are built
for the head size of every BASELINE configuration, d = 64 (VLMo-base / -large, ALBEF); the test-sized models have
narrower heads, which are zero-padded to 64 here (zero dimensions add nothing to q k^T and produce zero output
columns; the softmax scale stays d ** -0.5 of the true width); wider heads or another dtype on the GPU raise
``HipExtensionError``.  Host tensors -- the white boxes also run on the CPU, as
the frozen model behind the oracle in ``tests/`` -- go through PyTorch's ``scaled_dot_product_attention``.
"""
import torch
import torch.nn.functional as F

from .. import attention as _attn


def _on_gpu(t):
    """True: the HIP kernels run.  A GPU tensor they cannot take is an error, never a silent library fallback."""
    if not t.is_cuda:
        return False
    if t.dtype != torch.float32:
        raise _attn._hip.HipExtensionError("the white boxes' attention runs in fp32 on csrc/attn.hip, got {}".format(t.dtype))
    if t.shape[-1] > _attn.HEAD_DIM:
        raise _attn._hip.HipExtensionError(
            "csrc/attn.hip is built for heads of at most {} dimensions (every BASELINE configuration has 64), got {}; "
            "there is no library fallback on the GPU".format(_attn.HEAD_DIM, t.shape[-1]))
    return True


def _pad(t):
    return F.pad(t, (0, _attn.HEAD_DIM - t.shape[-1]))


def mha(q, k, v, bias=None):
    """softmax(q k^T / sqrt(d) + bias) v; q (B, Sq, H, d), k / v (B, Sk, H, d), bias broadcastable to
    (B, H, Sq, Sk) -> (B, Sq, H, d)."""
    d = q.shape[-1]
    if _on_gpu(q):
        if d == _attn.HEAD_DIM:
            return _attn.attention(q, k, v, bias)
        return _attn.attention(_pad(q), _pad(k), _pad(v), bias, scale=d ** -0.5)[..., :d]
    if isinstance(bias, _attn.KeyHoleBias):
        bias = bias.dense()
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias)
    return o.transpose(1, 2)


def mha_packed(qkv, bias=None):
    """Self-attention on a packed projection output qkv (B, S, 3, H, d) -> (B, S, H, d)."""
    d = qkv.shape[-1]
    if _on_gpu(qkv):
        if d == _attn.HEAD_DIM:
            return _attn.self_attention_packed(qkv, bias)
        return _attn.self_attention_packed(_pad(qkv), bias, scale=d ** -0.5)[..., :d]
    return mha(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias)
