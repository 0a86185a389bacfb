"""fp32 MFMA attention (csrc/attn.hip) against PyTorch's fp32 scaled_dot_product_attention on the same device.

Tolerance: both sides are fp32; the kernel's products are exact fp32 fma chains in a different order than the
library's, exp() is the hardware exponential (~2 ulp): 2e-5 absolute on outputs of O(1), 1e-4 relative on gradients.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(b, h, sq, sk, seed, packed=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    if packed:                                  # q, k, v as strided views of one (B, S, 3, H, 64) projection output
        qkv = torch.randn(b, sq, 3, h, 64, device=DEV, generator=g)
        return qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    return (torch.randn(b, sq, h, 64, device=DEV, generator=g), torch.randn(b, sk, h, 64, device=DEV, generator=g),
            torch.randn(b, sk, h, 64, device=DEV, generator=g))


def _sdpa(q, k, v, bias, scale=None):
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias,
                                       scale=scale)
    return o.transpose(1, 2)


@pytest.mark.parametrize("b,h,sq,sk,packed", [(2, 3, 77, 77, False), (1, 2, 587, 587, True), (2, 1, 33, 160, False),
                                               (3, 4, 128, 128, True), (1, 1, 1, 5, False), (1, 12, 617, 617, True),
                                               (1, 12, 25, 577, False)])     # ALBEF's cross-attention at batch 1
@pytest.mark.parametrize("bias_kind", ["none", "shared", "per_batch_padding"])
def test_attention_forward_matches_sdpa(b, h, sq, sk, packed, bias_kind):
    from vqattack_amd import attention
    if packed and sq != sk:
        pytest.skip("packed qkv is self-attention")
    q, k, v = _inputs(b, h, sq, sk, 1, packed)
    bias = None
    if bias_kind == "shared":                   # relative-position bias shared over the batch (stride 0)
        bias = (torch.randn(1, h, sq, sk, device=DEV) * 0.5).expand(b, -1, -1, -1)
    elif bias_kind == "per_batch_padding":      # -inf key padding per sample + a bias, materialised per batch
        bias = torch.randn(b, h, sq, sk, device=DEV) * 0.5
        for s in range(b):
            n_pad = (s * 3 + 1) % max(sk - 1, 1)
            if n_pad:
                bias[s, :, :, sk - n_pad:] = float("-inf")
    got, lse = attention.attention_forward(q, k, v, bias)
    want = _sdpa(q, k, v, bias)
    assert got.shape == want.shape
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())
    scores = torch.einsum("bqhd,bkhd->bhqk", q, k) * 64 ** -0.5
    if bias is not None:
        scores = scores + bias
    assert torch.allclose(lse, torch.logsumexp(scores, dim=-1), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("b,h,sq,sk", [(2, 3, 77, 77), (1, 2, 587, 587), (2, 1, 33, 160), (3, 4, 128, 128), (1, 1, 1, 5),
                                       (1, 12, 25, 577)])
@pytest.mark.parametrize("bias_kind", ["none", "shared", "per_batch_padding"])
@pytest.mark.parametrize("form", ["saved_scores", "ds_workspace", "recompute"])
def test_attention_backward_matches_sdpa(b, h, sq, sk, bias_kind, form, monkeypatch):
    from vqattack_amd import attention
    if form != "saved_scores":                  # the forward keeps no scores: 5 products with the dS workspace ...
        monkeypatch.setattr(attention, "SCORES_LIMIT", 0)
    if form == "recompute":                     # ... or the 7-product form that needs no workspace at all
        monkeypatch.setattr(attention, "DS_WORKSPACE_LIMIT", 0)
    q, k, v = (t.clone().requires_grad_(True) for t in _inputs(b, h, sq, sk, 2))
    bias = None
    if bias_kind == "shared":
        bias = (torch.randn(1, h, sq, sk, device=DEV) * 0.5).expand(b, -1, -1, -1)
    elif bias_kind == "per_batch_padding":
        bias = torch.randn(b, h, sq, sk, device=DEV) * 0.5
        for s in range(b):
            n_pad = (s * 3 + 1) % max(sk - 1, 1)
            if n_pad:
                bias[s, :, :, sk - n_pad:] = float("-inf")
    go = torch.randn(b, sq, h, 64, device=DEV)
    out = attention.attention(q, k, v, bias)
    out.backward(go)
    got = [t.grad.clone() for t in (q, k, v)]
    for t in (q, k, v):
        t.grad = None
    _sdpa(q, k, v, bias).backward(go)
    for name, g, t in zip("qkv", got, (q, k, v)):
        scale = float(t.grad.abs().max()) + 1e-6
        assert float((g - t.grad).abs().max()) <= 1e-4 * scale, (name, float((g - t.grad).abs().max()), scale)


def test_packed_self_attention_gradient_and_determinism():
    from vqattack_amd import attention
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(2, 200, 3, 4, 64, device=DEV, generator=g, requires_grad=True)
    bias = (torch.randn(1, 4, 200, 208, device=DEV, generator=g) * 0.3)[..., :200].expand(2, -1, -1, -1)
    go = torch.randn(2, 200, 4, 64, device=DEV, generator=g)
    grads = []
    for _ in range(3):
        o = attention.self_attention_packed(qkv, bias)
        o.backward(go)
        grads.append(qkv.grad.clone())
        qkv.grad = None
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])      # no atomics: bitwise reproducible
    ref = _sdpa(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias)
    assert torch.allclose(o, ref, atol=2e-5, rtol=1e-5)
    ref.backward(go)
    assert float((grads[0] - qkv.grad).abs().max()) <= 1e-4 * float(qkv.grad.abs().max())


@pytest.mark.parametrize("b,h,s", [(3, 2, 77), (4, 12, 591), (2, 1, 40)])
def test_key_hole_equals_the_dense_padding_mask(b, h, s):
    """``KeyHoleBias``: one (1, H, S, S) slab shared by the batch + a per-sample masked key range [lo, hi) applied inside
    the forward kernel -- against the same mask written into a per-sample dense bias, forward and backward, packed qkv;
    holes inside the first key tile, across the first tile boundary, empty, and at the very end of the sequence."""
    from vqattack_amd import attention
    g = torch.Generator(device=DEV).manual_seed(s)
    qkv = torch.randn(b, s, 3, h, 64, device=DEV, generator=g)
    slab = torch.zeros(1, h, s, (s + 31) // 32 * 32, device=DEV)
    slab[..., :s] = torch.randn(1, h, s, s, device=DEV, generator=g) * 0.5
    slab = slab[..., :s]
    holes = [[5, 14], [30, 37], [9, 9], [s - 3, s]][:b]
    hole = torch.tensor(holes, dtype=torch.int32, device=DEV)
    khb = attention.KeyHoleBias(slab.expand(b, -1, -1, -1), hole)
    dense = khb.dense()
    assert dense.shape == (b, h, s, s) and bool(torch.isinf(dense[0, 0, 0, 5:14]).all()) and bool(
        torch.isfinite(dense[0, 0, 0, 14:]).all())
    go = torch.randn(b, s, h, 64, device=DEV, generator=g)
    a = qkv.clone().requires_grad_(True)
    out_hole = attention.self_attention_packed(a, khb)
    out_hole.backward(go)
    c = qkv.clone().requires_grad_(True)
    out_dense = attention.self_attention_packed(c, dense)
    out_dense.backward(go)
    assert torch.equal(out_hole, out_dense)                   # the same arithmetic: -inf enters the same accumulator
    assert torch.equal(a.grad, c.grad)
    ref_in = qkv.clone().requires_grad_(True)
    ref = _sdpa(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], dense)
    ref.backward(go)
    assert torch.allclose(out_hole, ref, atol=2e-5, rtol=1e-5)
    assert float((a.grad - ref_in.grad).abs().max()) <= 1e-4 * float(ref_in.grad.abs().max())
    # masked keys receive no gradient
    for i, (lo, hi) in enumerate(holes):
        if hi > lo:
            assert float(a.grad[i, lo:hi, 1:].abs().max()) == 0.0        # dK, dV rows of the hole
    with torch.no_grad():                                      # forward-only path (targets, black-box scoring)
        assert torch.equal(attention.self_attention_packed(qkv, khb), out_hole.detach())


# ---- the reference's own resolution: 480 x 480 images = 30 x 30 + 1 = 901 image tokens (ALBEF_attack/configs/VQA.yaml:10,
# vlmo/config.py:283-299).  VLMo: 40 text + 901 = 941 tokens, 915 after the padding trim of a 12-word batch; ALBEF's
# fusion layers: <= 25 text queries over the 901 image keys (multiway_transformer.py:88-118, xbert.py cross attention).
# More key tiles than any 384-px case (30 instead of 19-20), a ragged last tile (941 = 29 * 32 + 13), the lazy-max
# rescale over a longer row, and saved scores / dS^T workspaces at (8 * 128) x (30 * 32) per (batch, head).
@pytest.mark.parametrize("s", [915, 941])
@pytest.mark.parametrize("form", ["saved_scores", "ds_workspace", "recompute"])
def test_self_attention_at_the_480px_token_counts(s, form, monkeypatch):
    from vqattack_amd import attention
    if form != "saved_scores":
        monkeypatch.setattr(attention, "SCORES_LIMIT", 0)
    if form == "recompute":
        monkeypatch.setattr(attention, "DS_WORKSPACE_LIMIT", 0)
    b, h = 3, 12
    g = torch.Generator(device=DEV).manual_seed(s)
    qkv = torch.randn(b, s, 3, h, 64, device=DEV, generator=g)
    slab = torch.zeros(1, h, s, (s + 31) // 32 * 32, device=DEV)
    slab[..., :s] = torch.randn(1, h, s, s, device=DEV, generator=g) * 0.5
    slab = slab[..., :s]
    n_text = s - 901                                        # padded text keys of sample i: [n_i, n_text)
    holes = [[6, n_text], [n_text, n_text], [n_text - 1, n_text]]
    khb = attention.KeyHoleBias(slab.expand(b, -1, -1, -1), torch.tensor(holes, dtype=torch.int32, device=DEV))
    dense = khb.dense()
    go = torch.randn(b, s, h, 64, device=DEV, generator=g)
    grads, outs = [], []
    for _ in range(2):
        a = qkv.clone().requires_grad_(True)
        out = attention.self_attention_packed(a, khb)
        out.backward(go)
        grads.append(a.grad)
        outs.append(out.detach())
    assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])          # bitwise reproducible
    ref_in = qkv.clone().requires_grad_(True)
    ref = _sdpa(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], dense)
    ref.backward(go)
    assert torch.allclose(outs[0], ref, atol=2e-5, rtol=1e-5), float((outs[0] - ref).abs().max())
    err = float((grads[0] - ref_in.grad).abs().max())
    assert err <= 1e-4 * float(ref_in.grad.abs().max()), (err, float(ref_in.grad.abs().max()))
    assert float(grads[0][0, 6:n_text, 1:].abs().max()) == 0.0                      # masked keys: no dK / dV


@pytest.mark.parametrize("bias_kind", ["none", "shared"])
@pytest.mark.parametrize("form", ["saved_scores", "ds_workspace", "recompute"])
def test_cross_attention_25_text_queries_over_901_image_keys(bias_kind, form, monkeypatch):
    from vqattack_amd import attention
    if form != "saved_scores":
        monkeypatch.setattr(attention, "SCORES_LIMIT", 0)
    if form == "recompute":
        monkeypatch.setattr(attention, "DS_WORKSPACE_LIMIT", 0)
    b, h, sq, sk = 2, 12, 25, 901
    q, k, v = (t.clone().requires_grad_(True) for t in _inputs(b, h, sq, sk, 11))
    bias = (torch.randn(1, h, sq, sk, device=DEV) * 0.5).expand(b, -1, -1, -1) if bias_kind == "shared" else None
    go = torch.randn(b, sq, h, 64, device=DEV)
    out = attention.attention(q, k, v, bias)
    out.backward(go)
    got = [t.grad.clone() for t in (q, k, v)]
    for t in (q, k, v):
        t.grad = None
    ref = _sdpa(q, k, v, bias)
    ref.backward(go)
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5), float((out - ref).abs().max())
    for name, g_, t in zip("qkv", got, (q, k, v)):
        scale = float(t.grad.abs().max()) + 1e-6
        assert float((g_ - t.grad).abs().max()) <= 1e-4 * scale, (name, float((g_ - t.grad).abs().max()), scale)


@pytest.mark.parametrize("b,h,s,n", [(1, 12, 591, 8), (1, 12, 591, 3), (2, 3, 200, 2), (1, 2, 915, 5), (1, 1, 64, 2)])
def test_loop_split_equals_the_unsplit_kernels(b, h, s, n, monkeypatch):
    """Small batches cut the kernels' tile loops into ``n`` parts (workgroups per (batch, head, block) instead of one) and
    combine the partial results in part order: forward (flash-decoding reduction of partial maxima / sums / accumulators)
    and the scores-based backward (partial dk / dv / dq summed) against the unsplit kernels on the same operands -- with
    a shared bias slab and a key hole that crosses a part boundary -- and bitwise reproducible run to run."""
    from vqattack_amd import attention
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = torch.randn(b, s, 3, h, 64, device=DEV, generator=g)
    pad = (s + 31) // 32 * 32
    slab = (torch.randn(1, h, s, pad, device=DEV, generator=g) * 0.4)[..., :s].expand(b, -1, -1, -1)
    hole = torch.tensor([[min(30 + 7 * i, s - 2), min(70 + 9 * i, s - 1)] for i in range(b)], dtype=torch.int32, device=DEV)
    go = torch.randn(b, s, h, 64, device=DEV, generator=g)

    def run(parts):
        monkeypatch.setenv("VQA_ATTN_SPLIT", str(parts))
        assert attention.loop_split(b, h, s, s, DEV) == parts
        x = qkv.clone().requires_grad_(True)
        o = attention.self_attention_packed(x, attention.KeyHoleBias(slab, hole))
        o.backward(go)
        return o.detach(), x.grad

    o1, g1 = run(1)
    on, gn = run(n)
    on2, gn2 = run(n)
    assert torch.equal(on, on2) and torch.equal(gn, gn2)                       # fixed part order: no atomics
    assert torch.allclose(on, o1, atol=2e-6, rtol=1e-5), float((on - o1).abs().max())
    assert float((gn - g1).abs().max()) <= 2e-5 * float(g1.abs().max())
    ref = _sdpa(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], attention.KeyHoleBias(slab, hole).dense())
    assert torch.allclose(on, ref, atol=2e-5, rtol=1e-5)


def test_loop_split_heuristic_and_abi_validation(monkeypatch):
    from vqattack_amd import _hip, attention
    monkeypatch.delenv("VQA_ATTN_SPLIT", raising=False)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert attention.loop_split(64, 12, 591, 591, DEV) == 1                  # the benchmark's shape fills the chip
    n1 = attention.loop_split(1, 12, 591, 591, DEV)                          # the reference's own batch 1 does not
    assert 2 <= n1 <= 8 and 12 * 5 * n1 <= 2 * cus + 12 * 5
    assert attention.loop_split(1, 1, 20, 20, DEV) == 1                      # one tile: nothing to cut
    assert attention.loop_split(1, 12, 25, 577, DEV) == 1                    # cross-attention, backward: one query tile
    assert attention.loop_split(1, 12, 25, 577, DEV, backward=False) >= 4    # ... its forward loops over 19 key tiles
    lib = _hip.lib()
    assert lib.vqa_attn_split_ws_floats(1, 12, 591, 591, 1) == 0
    assert lib.vqa_attn_split_ws_floats(1, 12, 591, 591, 4) == max(12 * 591 * 4 * 66, 2 * 12 * 591 * 4 * 64)
    q = torch.randn(1, 40, 2, 64, device=DEV)
    o, lse = torch.empty_like(q), torch.empty(1, 2, 40, device=DEV)
    strides = attention._longs([q.stride(0), q.stride(1), q.stride(2)] * 4)
    ws = torch.empty(lib.vqa_attn_split_ws_floats(1, 2, 40, 40, 2), device=DEV)
    args = (_hip.ptr(q), _hip.ptr(q), _hip.ptr(q), None, _hip.ptr(o), _hip.ptr(lse), None, 1, 2, 40, 40, strides, None, 0.125, None)
    assert lib.vqa_attn_fwd(*args, 2, _hip.ptr(ws), None) == 0
    assert lib.vqa_attn_fwd(*args, 3, _hip.ptr(ws), None) == -2                       # VQA_ERR_SHAPE: 2 key tiles: at most 2 parts
    assert lib.vqa_attn_fwd(*args, 2, None, None) == -1                              # VQA_ERR_NULL
    assert lib.vqa_attn_fwd(*args, 0, None, None) == -2
    torch.cuda.synchronize()
