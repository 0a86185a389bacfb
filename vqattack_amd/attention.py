"""fp32 attention of the frozen white boxes on the gfx950 matrix cores (``csrc/attn.hip``), with autograd.

``attention(q, k, v, bias, scale)``: ``q`` (B, Sq, H, 64), ``k`` / ``v`` (B, Sk, H, 64) -- any strides with a dense head
dimension, e.g. the three slices of a packed ``qkv`` projection -- and an optional additive ``bias`` broadcastable to
(B, H, Sq, Sk) with a dense last dimension (relative-position bias and / or ``-inf`` key padding).  Returns
``softmax(scale * q k^T + bias) v`` as (B, Sq, H, 64) contiguous, so the ``reshape(B, Sq, H * 64)`` that follows in a
transformer block is free.  The white boxes' ``Attention.forward`` (reference: ``vlmo/modules/multiway_transformer.py:
88-118``) is a callee of the attack's hot path; this replaces ``F.scaled_dot_product_attention`` there (same math, exact
fp32 products on ``v_mfma_f32_32x32x2_f32``).  There is no CPU path.

Backward: deterministic (no float atomics).  A forward whose inputs require a gradient also saves its pre-softmax scores
(``SCORES_LIMIT``), and the backward keeps dS in a transient workspace (``DS_WORKSPACE_LIMIT``): 4 matrix products
instead of the 7 of the workspace-free form, which remains the fallback above the limits.  The bias is frozen (no
gradient).  A query row whose keys are all masked with ``-inf`` yields NaN, as the library's attention does.
"""
import ctypes

import torch

from . import _hip
from ._hip import check, lib, ptr, stream_for

HEAD_DIM = 64


def _bshd(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[-1] != HEAD_DIM:
        raise _hip.HipExtensionError("{} must be a float32 (B, S, H, 64) HIP tensor, got {} {}".format(
            name, t.dtype, tuple(t.shape)))
    if t.stride(-1) != 1 or any(s % 4 for s in t.stride()[:-1]) or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _bias_view(bias, b, h, sq, sk):
    """(tensor kept alive, strides (batch, head, row)) of a bias broadcastable to (B, H, Sq, Sk).

    The kernels read a bias row in whole tiles of 32 keys with 16-byte loads, the last, partial tile included (what
    lies beyond Sk is masked, not used): from every row start, ceil32(Sk) floats must be readable.  A bias whose
    storage does not guarantee that -- or that is not 16-byte aligned -- is copied once, in its own un-broadcast shape,
    into rows padded to a multiple of 32 floats; producers that build their bias that way (``FrozenVlmo.attention_bias``)
    are used in place."""
    if bias is None:
        return None, None
    if bias.dtype != torch.float32 or not bias.is_cuda:
        raise TypeError("bias must be a float32 HIP tensor")
    bias, in_place = _bias_layout(bias, sk)
    if not in_place:
        store = torch.zeros(tuple(bias.shape[:-1]) + ((sk + 31) // 32 * 32,), dtype=torch.float32, device=bias.device)
        store[..., :sk] = bias
        bias = store[..., :sk]
    bias = bias.expand(b, h, sq, sk)
    return bias, (bias.stride(0), bias.stride(1), bias.stride(2))


def _bias_layout(bias, sk):
    """(bias as 4-d with its broadcast dimensions collapsed to size 1, whether the kernels can read it in place).

    In place needs: unit key stride, 16-byte aligned base, every non-broadcast leading stride a non-negative multiple
    of 4 floats, and ceil32(Sk) floats inside the storage from the start of the LAST row (rows are consecutive in
    memory, so every earlier row then has them too).  Device-independent: checked on CPU tensors in
    tests/test_host_logic.py."""
    while bias.dim() < 4:
        bias = bias.unsqueeze(0)
    if bias.dim() != 4 or bias.shape[-1] != sk:
        raise ValueError("bias must be broadcastable to (B, H, Sq, Sk = {}), got {}".format(sk, tuple(bias.shape)))
    # broadcast (stride-0) dimensions are collapsed first, so that a copy, if one is needed, stays small
    bias = bias[tuple(slice(0, 1) if st == 0 else slice(None) for st in bias.stride()[:-1])]
    pad = (sk + 31) // 32 * 32
    last_row = bias.storage_offset() + sum((n - 1) * st for n, st in zip(bias.shape[:-1], bias.stride()[:-1]))
    room = last_row + pad <= bias.untyped_storage().nbytes() // 4
    lead = [st for st, n in zip(bias.stride()[:-1], bias.shape[:-1]) if n > 1]
    in_place = (bias.stride(-1) == 1 and room and not any(st % 4 or st < 0 for st in lead)
                and bias.data_ptr() % 16 == 0)
    return bias, in_place


def _longs(vals):
    return (ctypes.c_long * len(vals))(*[int(v) for v in vals])


class KeyHoleBias:
    """An additive attention mask in two parts: ``slab`` -- a bias broadcastable to (B, H, Sq, Sk), typically ONE
    (1, H, S, S) relative-position slab shared by the batch -- and ``hole``, int32 (B, 2): keys ``[hole[b, 0], hole[b, 1])``
    of sample b are masked for every head and query (the padded text tokens of a question shorter than the batch's text
    length).  The forward kernel applies the hole itself (``key_hole`` of ``vqa_attn_fwd``), so a ragged batch does not
    need a per-sample (B, H, S, S) copy of the slab; ``dense()`` is that copy, for consumers that want a plain tensor."""

    def __init__(self, slab, hole):
        self.slab, self.hole = slab, hole

    def detach(self):
        return self

    def dense(self):
        sk = self.slab.shape[-1]
        b = self.hole.shape[0]
        keys = torch.arange(sk, device=self.slab.device)
        masked = (keys[None, :] >= self.hole[:, :1]) & (keys[None, :] < self.hole[:, 1:2])          # (B, Sk)
        pad = torch.zeros(b, 1, 1, sk, device=self.slab.device).masked_fill(masked[:, None, None, :], float("-inf"))
        return self.slab + pad


def _split_hole(bias):
    """(tensor bias or None, key hole or None) of a bias argument."""
    if isinstance(bias, KeyHoleBias):
        return bias.slab, bias.hole
    return bias, None


_CUS = {}


def loop_split(b, h, sq, sk, device, backward=True):
    """Parts the kernels' tile loops are cut into for this shape (1 = not at all).  A launch has
    ``b * h * ceil(s / 128)`` workgroups, each a serial chain over ``ceil(s / 32)`` tiles: with fewer workgroups than the
    chip can hold (2 per CU) -- the reference's own batch 1: 60 on 256 CUs -- the loop is cut so that about that many
    run, at most 8 parts and never fewer than 2 tiles per part.  The forward loops over KEY tiles; the backward's two
    kernels loop over query tiles (dK / dV) and key tiles (dQ), so its count is bounded by the shorter side (ALBEF's
    cross-attention: 25 text queries on 577 image keys splits forward only).  ``VQA_ATTN_SPLIT=n`` pins it (experiments)."""
    import os
    pinned = os.environ.get("VQA_ATTN_SPLIT")
    tiles = ((min(sq, sk) if backward else sk) + 31) // 32
    if pinned:
        return max(1, min(int(pinned), tiles))
    dev = torch.device(device).index or 0
    if dev not in _CUS:
        _CUS[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    wgs = b * h * ((max(sq, sk) + 127) // 128)
    want = 2 * _CUS[dev]
    if wgs * 2 > want:
        return 1
    return max(1, min(8, tiles // 2, -(-want // wgs)))


def _split_ws(b, h, sq, sk, nsplit, device):
    if nsplit <= 1:
        return None
    return torch.empty(int(lib().vqa_attn_split_ws_floats(b, h, sq, sk, nsplit)), dtype=torch.float32, device=device)


def _forward(q, k, v, bias, bstr, scale, save_scores=False, key_hole=None):
    """Returns (o, lse, scores); ``scores`` is None unless ``save_scores`` and the buffer fits SCORES_LIMIT.
    ``key_hole``: int32 (B, 2) device tensor, see ``KeyHoleBias``."""
    b, sq, h, _ = q.shape
    sk = k.shape[1]
    nsplit = loop_split(b, h, sq, sk, q.device, backward=False)
    sws = _split_ws(b, h, sq, sk, nsplit, q.device)
    o = torch.empty((b, sq, h, HEAD_DIM), dtype=torch.float32, device=q.device)
    lse = torch.empty((b, h, sq), dtype=torch.float32, device=q.device)
    scores = None
    if save_scores:
        if scores_fit(b, h, sq, sk):
            scores = torch.empty(int(lib().vqa_attn_scores_floats(b, h, sq, sk)), dtype=torch.float32, device=q.device)
        elif key_hole is not None:
            raise _hip.HipExtensionError("a key hole needs the saved-scores backward (the recomputing forms read the "
                                         "padding from the bias): pass KeyHoleBias.dense() at this size")
    strides = _longs([q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                      v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2)])
    if key_hole is not None and (key_hole.dtype != torch.int32 or tuple(key_hole.shape) != (b, 2) or
                                 not key_hole.is_cuda or not key_hole.is_contiguous()):
        raise TypeError("key_hole must be a contiguous int32 (B, 2) HIP tensor")
    with torch.cuda.device(q.device):
        check(lib().vqa_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(bias), ptr(o), ptr(lse), ptr(scores), b, h, sq, sk, strides,
                                 _longs(bstr) if bstr else None, scale,
                                 None if key_hole is None else ctypes.c_void_p(key_hole.data_ptr()), nsplit, ptr(sws),
                                 stream_for(q)),
              "vqa_attn_fwd")
    return o, lse, scores


def scores_fit(b, h, sq, sk):
    """True when a differentiated forward of this shape saves its scores (and the backward gets its dS workspace)."""
    n = int(lib().vqa_attn_scores_floats(b, h, sq, sk))
    return 0 < 4 * n <= SCORES_LIMIT and 0 < 4 * int(lib().vqa_attn_bwd_ws_floats(b, h, sq, sk)) <= DS_WORKSPACE_LIMIT


def _prepare(q, k, v, bias, scale, need_grad=False):
    """Returns (q, k, v, bias tensor, bias strides, scale, key hole).  A ``KeyHoleBias`` keeps its two-part form when the
    kernels can honour it (no gradient needed, or the saved-scores backward applies); otherwise it is densified."""
    q, k, v = _bshd(q, "q"), _bshd(k, "k"), _bshd(v, "v")
    b, sq, h, _ = q.shape
    sk = k.shape[1]
    if k.shape != (b, sk, h, HEAD_DIM) or v.shape != k.shape:
        raise ValueError("q / k / v shape mismatch: {} {} {}".format(tuple(q.shape), tuple(k.shape), tuple(v.shape)))
    hole = None
    if isinstance(bias, KeyHoleBias):
        if need_grad and not scores_fit(b, h, sq, sk):
            bias = bias.dense()
        else:
            bias, hole = bias.slab, bias.hole
    bias, bstr = _bias_view(bias, b, h, sq, sk)
    return q, k, v, bias, bstr, HEAD_DIM ** -0.5 if scale is None else float(scale), hole


def attention_forward(q, k, v, bias=None, scale=None):
    """Forward only; returns ``(o (B, Sq, H, 64), lse (B, H, Sq))``."""
    q, k, v, bias, bstr, scale, hole = _prepare(q, k, v, bias, scale)
    return _forward(q, k, v, bias, bstr, scale, key_hole=hole)[:2]


# The backward keeps dS in a transient workspace (5 matrix products) when that workspace is at most this many bytes,
# and recomputes the scores in both of its kernels (7 products, no workspace) above it.
DS_WORKSPACE_LIMIT = 8 << 30
# A forward that will be differentiated also saves its pre-softmax scores for the backward (4 products) when they are
# at most this many bytes per call; they live from the forward to the backward of the same layer (12-24 layers deep:
# 14 GB for VLMO-base at batch 64, 58 GB for ALBEF-base at batch 256, of the 288 GB of HBM).
SCORES_LIMIT = 6 << 30


def _backward(q, k, v, bias, bstr, o, lse, go, dq, dk, dv, scale, workspace=True, scores=None):
    b, sq, h, _ = q.shape
    sk = k.shape[1]
    if go.stride(-1) != 1 or any(s % 4 for s in go.stride()[:-1]) or go.data_ptr() % 16:
        go = go.contiguous()
    delta = torch.empty((b, h, sq), dtype=torch.float32, device=q.device)
    ws = None
    if workspace:
        n = int(lib().vqa_attn_bwd_ws_floats(b, h, sq, sk))
        if 0 < 4 * n <= DS_WORKSPACE_LIMIT:
            ws = torch.empty(n, dtype=torch.float32, device=q.device)
    strides = _longs([q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                      v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2)])
    gstr = _longs([go.stride(0), go.stride(1), go.stride(2), dq.stride(0), dq.stride(1), dq.stride(2),
                   dk.stride(0), dk.stride(1), dk.stride(2), dv.stride(0), dv.stride(1), dv.stride(2)])
    nsplit = loop_split(b, h, sq, sk, q.device) if (ws is not None and scores is not None) else 1
    sws = _split_ws(b, h, sq, sk, nsplit, q.device)
    with torch.cuda.device(q.device):
        check(lib().vqa_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(bias), ptr(o), ptr(go), ptr(lse),
                                 ptr(scores if ws is not None else None), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ptr(ws),
                                 b, h, sq, sk, strides, _longs(bstr) if bstr else None, gstr, scale, nsplit, ptr(sws),
                                 stream_for(q)),
              "vqa_attn_bwd")


class _Attention(torch.autograd.Function):
    """Separate q, k, v (B, S, H, 64); gradients come back as three contiguous tensors."""

    @staticmethod
    def forward(ctx, q, k, v, bias, scale):
        need = any(ctx.needs_input_grad[:3])
        q, k, v, bias_t, bstr, scale, hole = _prepare(q.detach(), k.detach(), v.detach(),
                                                      None if bias is None else bias.detach(), scale, need_grad=need)
        o, lse, scores = _forward(q, k, v, bias_t, bstr, scale, save_scores=need, key_hole=hole)
        ctx.save_for_backward(q, k, v, o, lse, bias_t, scores)
        ctx.bstr, ctx.scale = bstr, scale
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, lse, bias, scores = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q, memory_format=torch.contiguous_format), \
            torch.empty_like(k, memory_format=torch.contiguous_format), \
            torch.empty_like(v, memory_format=torch.contiguous_format)
        _backward(q, k, v, bias, ctx.bstr, o, lse, go, dq, dk, dv, ctx.scale, scores=scores)
        return dq, dk, dv, None, None


class _PackedSelfAttention(torch.autograd.Function):
    """Packed projection output qkv (B, S, 3, H, 64): the gradient is ONE packed buffer the kernels write in place --
    no select / stack copies around the attention in the backward pass."""

    @staticmethod
    def forward(ctx, qkv, bias, scale):
        qkv = qkv.detach()
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        need = ctx.needs_input_grad[0]
        q, k, v, bias_t, bstr, scale, hole = _prepare(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2],
                                                      None if bias is None else bias.detach(), scale, need_grad=need)
        o, lse, scores = _forward(q, k, v, bias_t, bstr, scale, save_scores=need, key_hole=hole)
        ctx.save_for_backward(qkv, o, lse, bias_t, scores)
        ctx.bstr, ctx.scale = bstr, scale
        return o

    @staticmethod
    def backward(ctx, go):
        qkv, o, lse, bias, scores = ctx.saved_tensors
        dqkv = torch.empty_like(qkv)
        _backward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias, ctx.bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1],
                  dqkv[:, :, 2], ctx.scale, scores=scores)
        return dqkv, None, None


def attention(q, k, v, bias=None, scale=None):
    """``softmax(scale * q k^T + bias) v`` with autograd; q (B, Sq, H, 64), k / v (B, Sk, H, 64) -> (B, Sq, H, 64)."""
    return _Attention.apply(q, k, v, bias, scale)


def self_attention_packed(qkv, bias=None, scale=None):
    """Self-attention on a packed projection output ``qkv`` (B, S, 3, H, 64) -> (B, S, H, 64), with autograd."""
    if qkv.dim() != 5 or qkv.shape[2] != 3 or qkv.shape[-1] != HEAD_DIM or not qkv.is_cuda or qkv.dtype != torch.float32:
        raise _hip.HipExtensionError("qkv must be a float32 (B, S, 3, H, 64) HIP tensor")
    return _PackedSelfAttention.apply(qkv, bias, scale)
