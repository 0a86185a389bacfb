"""The fused block glue (``csrc/block.hip``) and the graph-free encoder built on it (``whitebox/_fused.py``) against plain
PyTorch fp32 on the same device.

Reference computation: ``Block.forward`` (``VLMO_VQAttack/vlmo/modules/multiway_transformer.py:184-201``: layer scale,
modality-expert LayerNorm / FFN split at the text length) and ALBEF's ViT block.  Tolerances: kernel outputs 2e-6
absolute on O(1) values (LayerNorm statistics in a different summation order than ATen's), gradients 1e-5 of the largest
entry; the whole encoder (12 layers deep) 2e-5 / 2e-4.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _split(t, b, s, n):
    """(B*S, D) -> the two contiguous per-modality buffers."""
    t3 = t.view(b, s, -1)
    return t3[:, :n].reshape(b * n, -1).contiguous(), t3[:, n:].reshape(b * (s - n), -1).contiguous()


def _merge(t0, t1, b, s, n):
    d = t0.shape[-1]
    return torch.cat([t0.view(b, n, d), t1.view(b, s - n, d)], dim=1).reshape(b * s, d)


@pytest.mark.parametrize("d", [64, 768, 1024, 200])
@pytest.mark.parametrize("mode", ["plain", "residual", "residual_scaled_split_in", "split_out", "both_split"])
def test_ln_fwd_and_bwd_match_torch(d, mode):
    from vqattack_amd import ops
    b, s, n = 3, 11, 4
    rows = b * s
    g = torch.Generator(device=DEV).manual_seed(d + len(mode))
    rnd = lambda *shape: torch.randn(*shape, device=DEV, generator=g)          # noqa: E731
    x, r = rnd(rows, d), rnd(rows, d)
    scale = rnd(d) * 0.5 + 1.0
    gam = [rnd(d) * 0.3 + 1.0, rnd(d) * 0.3 + 1.0]
    bet = [rnd(d) * 0.1, rnd(d) * 0.1]
    eps = 1e-6
    with_r = mode != "plain" and mode != "split_out"
    r_split = mode in ("residual_scaled_split_in", "both_split")
    y_split = mode in ("split_out", "both_split")
    use_scale = mode in ("residual_scaled_split_in", "both_split")
    # ---- torch reference
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True)
    xo = xr + (scale if use_scale else 1.0) * rr if with_r else xr
    if y_split:
        x3 = xo.view(b, s, d)
        want = torch.cat([F.layer_norm(x3[:, :n], (d,), gam[0], bet[0], eps),
                          F.layer_norm(x3[:, n:], (d,), gam[1], bet[1], eps)], dim=1).reshape(rows, d)
    else:
        want = F.layer_norm(xo, (d,), gam[0], bet[0], eps)
    dy, g_a, g_inj = rnd(rows, d), rnd(rows, d), rnd(rows, d)
    # dx of the kernel = gradient w.r.t. x_out with the two extra addends
    grads = torch.autograd.grad(want, [xo] if with_r else [xr], dy, retain_graph=with_r)
    want_dx = grads[0] + g_a + g_inj
    # ---- kernels
    y0 = torch.empty(b * n if y_split else rows, d, device=DEV)
    y1 = torch.empty(b * (s - n), d, device=DEV) if y_split else None
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    x_out = torch.empty_like(x) if with_r else None
    r0, r1 = (_split(r, b, s, n) if r_split else (r, None)) if with_r else (None, None)
    split_any = r_split or y_split
    ops.ln_fwd(x, gam[0], bet[0], y0, mean, rstd, eps, r0=r0, r1=r1, rscale=scale if use_scale else None, x_out=x_out,
               gamma1=gam[1] if y_split else None, beta1=bet[1] if y_split else None, y1=y1,
               period=s if split_any else 0, split=n if split_any else 0)
    got = _merge(y0, y1, b, s, n) if y_split else y0
    assert torch.allclose(got, want.detach(), rtol=0, atol=5e-6), float((got - want).abs().max())
    if with_r:
        assert torch.allclose(x_out, xo.detach(), rtol=0, atol=1e-6)
    ref_in = xo.detach() if with_r else x
    assert torch.allclose(mean, ref_in.mean(-1), atol=1e-6) and torch.allclose(
        rstd, (ref_in.var(-1, unbiased=False) + eps).rsqrt(), rtol=2e-6)
    dx = torch.empty_like(x)
    dr0 = torch.empty_like(r0) if with_r and (use_scale or r_split) else None
    dr1 = torch.empty_like(r1) if (with_r and r_split) else None
    dy0, dy1 = _split(dy, b, s, n) if y_split else (dy, None)
    ops.ln_bwd(dy0, x_out if with_r else x, mean, rstd, gam[0], dx, dy1=dy1, gamma1=gam[1] if y_split else None, g_a=g_a,
               g_inj=g_inj, rscale=scale if use_scale else None, dr0=dr0, dr1=dr1, period=s if split_any else 0,
               split=n if split_any else 0)
    tol = 1e-5 * float(want_dx.abs().max())
    assert float((dx - want_dx).abs().max()) <= tol
    if dr0 is not None:
        want_dr = (scale if use_scale else 1.0) * want_dx
        got_dr = _merge(dr0, dr1, b, s, n) if r_split else dr0
        assert float((got_dr - want_dr).abs().max()) <= 2 * tol


@pytest.mark.parametrize("n", [7, 4096, 64 * 577 * 3072 // 512 + 3])
def test_gelu_kernels_match_torch(n):
    from vqattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(n)
    h = torch.randn(n, device=DEV, generator=g) * 3
    h[:4] = torch.tensor([0.0, -0.0, 40.0, -40.0], device=DEV)[:min(4, n)]
    da = torch.randn(n, device=DEV, generator=g)
    hr = h.clone().requires_grad_(True)
    want = F.gelu(hr)
    want.backward(da)
    # erf is this kernel's own (packed-FMA minimax polynomials, < 1 ulp): it rounds differently from the library erff
    # torch calls; in the left tail 1 + erf cancels, so the absolute error is ~ulp(1) * |x| / 2 on either side
    assert torch.allclose(ops.gelu_fwd(h), want.detach(), rtol=2e-6, atol=1e-6)
    got = ops.gelu_bwd(h, da.clone())
    assert torch.allclose(got, hr.grad, rtol=1e-5, atol=2e-6)
    ref64 = F.gelu(h.double())                               # against fp64: the kernel is as accurate as torch's fp32
    err_k = float((ops.gelu_fwd(h).double() - ref64).abs().max())
    err_t = float((want.detach().double() - ref64).abs().max())
    assert err_k <= max(2.0 * err_t, 1e-6), (err_k, err_t)


def _vlmo_small():
    from vqattack_amd.whitebox.vlmo import VlmoConfig
    return VlmoConfig(dim=128, depth=4, heads=2, vlffn_start=3, image_size=32, patch=8, max_text_len=8, n_answers=7)


def _check_encoder(run, n_out, tol_out, tol_grad):
    """``run(fused) -> (list of output tensors, list of leaves)``: the fused and the eager encoder on the same inputs;
    outputs and every input gradient under random output gradients must agree."""
    outs_f, leaves_f = run(True)
    outs_e, leaves_e = run(False)
    assert len(outs_f) == len(outs_e) == n_out
    g = torch.Generator(device=DEV).manual_seed(5)
    grads = [torch.randn(o.shape, device=DEV, generator=g) for o in outs_e]
    for k, (a, b) in enumerate(zip(outs_f, outs_e)):
        assert float((a - b).detach().abs().max()) <= tol_out * max(1.0, float(b.detach().abs().max())), k
    torch.autograd.backward(outs_f, grads, inputs=leaves_f)
    torch.autograd.backward(outs_e, grads, inputs=leaves_e)
    for lf, le in zip(leaves_f, leaves_e):
        gmax = float(le.grad.abs().max())
        assert gmax > 0 and float((lf.grad - le.grad).abs().max()) <= tol_grad * gmax
    return outs_f


@pytest.mark.parametrize("ragged", [False, True])
def test_vlmo_fused_encoder_equals_eager_blocks(ragged):
    """Expert layers (text / image LayerNorm + FFN, split at the trimmed text length) followed by a VL-FFN layer, layer
    scale != 1, key padding inside the batch (per-sample masks), gradient w.r.t. image AND text embeddings."""
    from vqattack_amd.whitebox.vlmo import FrozenVlmo
    cfg = _vlmo_small()
    model = FrozenVlmo(cfg, seed=2).to(DEV)
    with torch.no_grad():
        for blk in model.blocks:
            blk.gamma_1.mul_(0.7), blk.gamma_2.mul_(1.3)
            for ln in (blk.norm1, blk.norm2_text, blk.norm2_imag):
                ln.weight.add_(torch.randn_like(ln.weight) * 0.2), ln.bias.add_(torch.randn_like(ln.bias) * 0.1)
    ids = torch.tensor([[101, 5, 6, 7, 102, 0, 0, 0], [101, 8, 9, 102, 0, 0, 0, 0], [101, 3, 4, 5, 6, 7, 102, 0]],
                       device=DEV)
    if not ragged:
        ids = ids[:1].repeat(3, 1)
    masks = (ids != 0).long()
    g = torch.Generator(device=DEV).manual_seed(1)
    image = torch.empty(3, 3, 32, 32, device=DEV).uniform_(-1, 1, generator=g)
    emb = model.text_embeddings(ids)

    def run(fused):
        model.fused_blocks = fused
        img = image.clone().requires_grad_(True)
        txt = emb.clone().requires_grad_(True)
        feats, states = model.encode(img, txt, masks)
        return feats[1:] + [states], [img, txt]
    _check_encoder(run, cfg.depth + 1, 2e-5, 2e-4)
    model.fused_blocks = True
    with torch.no_grad():                                    # the no-grad path (targets, black-box scoring) saves nothing
        f2, s2 = model.encode(image, emb, masks)
    model.fused_blocks = False
    with torch.no_grad():
        f3, s3 = model.encode(image, emb, masks)
    assert float((s2 - s3).abs().max()) <= 2e-5 * float(s3.abs().max())


def test_albef_vit_fused_encoder_equals_eager_blocks():
    from vqattack_amd.whitebox.albef import AlbefConfig, FrozenAlbef
    cfg = AlbefConfig(dim=128, vit_depth=3, bert_depth=3, fusion_layer=1, heads=2, patch=8, image_size=32, n_answers=5,
                      decoder_depth=1, k_test=3, mlm_probability=0.0)
    model = FrozenAlbef(cfg, seed=4).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    image = torch.empty(2, 3, 32, 32, device=DEV).uniform_(-1, 1, generator=g)

    def run(fused):
        model.fused_blocks = fused
        img = image.clone().requires_grad_(True)
        states, feats = model.visual_encoder(img)
        return feats[1:] + [states], [img]
    _check_encoder(run, cfg.vit_depth + 1, 2e-5, 2e-4)


@pytest.mark.parametrize("how", ["load_state_dict", "assign", "in_place", "data_copy"])
def test_fused_encoder_follows_weight_updates(how):
    """The cached fused spec holds packed COPIES of ALBEF's q / k / v weights (and aliases VLMo's): after a
    load_state_dict (copying or assign=True) or an in-place update of the weights the fused path must run on the NEW
    weights, like the eager blocks do."""
    from vqattack_amd.whitebox.albef import AlbefConfig, FrozenAlbef
    from vqattack_amd.whitebox.vlmo import FrozenVlmo
    acfg = AlbefConfig(dim=128, vit_depth=2, bert_depth=2, fusion_layer=1, heads=2, patch=8, image_size=32, n_answers=5,
                       decoder_depth=1, k_test=3, mlm_probability=0.0)
    g = torch.Generator(device=DEV).manual_seed(5)
    image = torch.empty(2, 3, 32, 32, device=DEV).uniform_(-1, 1, generator=g)
    ids = torch.tensor([[101, 5, 6, 7, 102, 0, 0, 0], [101, 8, 9, 102, 0, 0, 0, 0]], device=DEV)
    for make, fwd in ((lambda seed: FrozenAlbef(acfg, seed=seed).to(DEV), lambda m: m.visual_encoder(image)[0]),
                      (lambda seed: FrozenVlmo(_vlmo_small(), seed=seed).to(DEV),
                       lambda m: m.encode(image, m.text_embeddings(ids), (ids != 0).long())[1])):
        model, donor = make(4), make(9)
        with torch.no_grad():
            model.fused_blocks = True
            before = fwd(model).clone()
            if how == "load_state_dict":
                model.load_state_dict(donor.state_dict())
            elif how == "assign":
                model.load_state_dict(donor.state_dict(), assign=True)
            elif how == "in_place":
                for p, q in zip(model.parameters(), donor.parameters()):
                    p.copy_(q)
            else:           # writes through .data are invisible to the weights key (own version counter): documented,
                for p, q in zip(model.parameters(), donor.parameters()):     # and the model offers invalidate_fused()
                    p.data.copy_(q)
                model.invalidate_fused()
            fused = fwd(model)
            model.fused_blocks = False
            eager = fwd(model)
        assert float((fused - before).abs().max()) > 1e-2, "the donor's weights should change the output"
        assert float((fused - eager).abs().max()) <= 2e-5 * float(eager.abs().max()), \
            "the fused encoder still runs on the weights of before the update ({})".format(how)


def test_vlmo_base_fused_encoder_equals_eager_blocks_and_frees_its_activations():
    """BASELINE configs[1] shape (12 x 768, 587-token layout), batch 2; afterwards no activation of the call is alive
    (outputs saved through save_for_backward: no reference cycle through the graph)."""
    import gc
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, vlmo_base
    model = FrozenVlmo(vlmo_base(384), seed=0).to(DEV)
    ids = torch.zeros(2, 40, dtype=torch.long, device=DEV)
    ids[0, :6] = torch.tensor([101, 11, 12, 13, 14, 102], device=DEV)
    ids[1, :10] = torch.tensor([101, 21, 22, 23, 24, 25, 26, 27, 28, 102], device=DEV)
    masks = (ids != 0).long()
    g = torch.Generator(device=DEV).manual_seed(3)
    image = torch.empty(2, 3, 384, 384, device=DEV).uniform_(-1, 1, generator=g)
    emb = model.text_embeddings(ids)[:, :10]

    def run(fused):
        model.fused_blocks = fused
        img = image.clone().requires_grad_(True)
        feats, states = model.encode(img, emb, masks[:, :10])
        return feats[1:] + [states], [img]
    torch.cuda.synchronize()
    _check_encoder(run, 13, 5e-5, 5e-4)
    del run
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    model.fused_blocks = True
    for _ in range(3):
        img = image.clone().requires_grad_(True)
        feats, states = model.encode(img, emb, masks[:, :10])
        torch.autograd.backward([feats[5], states], [torch.ones_like(feats[5]), torch.ones_like(states)], inputs=[img])
        del feats, states, img
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20), "activations of finished calls are still allocated"


def test_block_glue_wrappers_refuse_short_or_foreign_operands():
    """The kernels index raw pointers from the row count: a short buffer must be an exception in the wrapper, never an
    out-of-bounds access on the device."""
    from vqattack_amd import ops
    rows, d = 12, 64
    x = torch.randn(rows, d, device=DEV)
    gam, bet = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y = torch.empty(rows, d, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.ln_fwd(x, gam, bet, y, mean, rstd, 1e-6)
    with pytest.raises(ValueError):
        ops.ln_fwd(x, gam, bet, y[:-1], mean, rstd, 1e-6)                        # short output
    with pytest.raises(ValueError):
        ops.ln_fwd(x, gam, bet, y, mean[:-1].contiguous(), rstd, 1e-6)           # short statistics
    with pytest.raises(ValueError):
        ops.ln_fwd(x, gam[:-4].contiguous(), bet, y, mean, rstd, 1e-6)           # parameter vector of another width
    with pytest.raises(TypeError):
        ops.ln_fwd(x, gam, bet, y.cpu(), mean, rstd, 1e-6)                       # host tensor
    with pytest.raises(ValueError):                                             # split output whose parts do not add up
        ops.ln_fwd(x, gam, bet, torch.empty(4, d, device=DEV), mean, rstd, 1e-6, gamma1=gam, beta1=bet,
                   y1=torch.empty(7, d, device=DEV), period=6, split=2)
    dx = torch.empty_like(x)
    with pytest.raises(ValueError):
        ops.ln_bwd(y, x, mean, rstd, gam, dx, g_a=x[:-1])                        # short residual gradient
    with pytest.raises(ValueError):
        ops.gelu_bwd(x, torch.randn(rows - 1, d, device=DEV))
