"""Kernel-level parity: HIP entry points (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT for the L-infinity family (adds, clamps, multiply by +-1/0 -- correctly rounded, same order as the
reference's eager chain).  The L2/L1 family and the cosine loss involve reductions whose summation order differs from
ATen's, so they are held to the fp32 tolerances written next to each assert (relative 2e-6 on per-sample norms,
which propagates to <= 1e-6 absolute on perturbations of magnitude <= 1; 1e-5 relative on the loss).
"""
import numpy as np
import pytest
import torch

from oracle import cleverhans_cpu as o

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from vqattack_amd import ops
    return ops


def _rand(shape, seed, lo=-1.0, hi=1.0):
    r = np.random.RandomState(seed)
    return torch.from_numpy(r.uniform(lo, hi, shape).astype(np.float32))


def _grad(shape, seed):
    r = np.random.RandomState(seed)
    g = r.standard_normal(shape).astype(np.float32)
    flat = g.reshape(-1)
    n = flat.size
    idx = r.choice(n, size=min(n, max(n // 50, 4)), replace=False)
    flat[idx[0::4]] = 0.0
    flat[idx[1::4]] = -0.0
    flat[idx[2::4]] = np.nan
    flat[idx[3::4]] = np.float32(1e-42)       # denormal: sign must still be +1
    return torch.from_numpy(g)


def _same_bits(a, b):
    a = a.detach().cpu().numpy().view(np.uint32)
    b = b.detach().cpu().numpy().view(np.uint32)
    return np.array_equal(a, b)


SHAPES = [(4, 3, 384, 384), (2, 3, 32, 32), (1, 3, 5, 7), (3, 1, 1, 1), (1, 1, 1, 2)]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
@pytest.mark.parametrize("eps,eps_iter", [(0.125, 0.01), (8 / 255, 2 / 255)])
def test_linf_step_bitexact(shape, eps, eps_iter):
    ops = _ops()
    x0 = _rand(shape, 1)
    x = torch.clamp(x0 + _rand(shape, 2, -eps, eps), -1, 1)
    g = _grad(shape, 3)
    want = o.pgd_tail_given_grad(x, g, x0, eps_iter, eps, np.inf, -1, 1)
    got = ops.linf_step(x.to(DEV), g.to(DEV), x0.to(DEV), eps_iter, eps, -1, 1)
    assert _same_bits(got, want)
    # without clipping
    want = o.pgd_tail_given_grad(x, g, x0, eps_iter, eps, np.inf)
    got = ops.linf_step(x.to(DEV), g.to(DEV), x0.to(DEV), eps_iter, eps, None, None)
    assert _same_bits(got, want)


def test_linf_step_nan_inputs_and_inplace():
    ops = _ops()
    shape = (2, 3, 16, 16)
    x0 = _rand(shape, 4)
    x = torch.clamp(x0 + _rand(shape, 5, -0.125, 0.125), -1, 1)
    x.view(-1)[7] = float("nan")
    x0.view(-1)[11] = float("nan")
    g = _grad(shape, 6)
    want = o.pgd_tail_given_grad(x, g, x0, 0.01, 0.125, np.inf, -1, 1)
    xd = x.to(DEV)
    got = ops.linf_step(xd, g.to(DEV), x0.to(DEV), 0.01, 0.125, -1, 1, out=xd)   # in place
    assert got.data_ptr() == xd.data_ptr()
    assert _same_bits(got, want)


def test_linf_unaligned_views_take_scalar_path():
    ops = _ops()
    n = 4099
    base_x, base_g, base_0 = _rand((n + 1,), 7), _grad((n + 1,), 8), _rand((n + 1,), 9)
    x, g, x0 = base_x[1:], base_g[1:], base_0[1:]          # 4-byte aligned only
    want = o.pgd_tail_given_grad(x.clone(), g.clone(), x0.clone(), 0.01, 0.125, np.inf, -1, 1)
    xd, gd, x0d = base_x.to(DEV)[1:], base_g.to(DEV)[1:], base_0.to(DEV)[1:]
    out = torch.empty(n + 1, device=DEV)[1:]
    from vqattack_amd import _hip
    _hip.check(_hip.lib().vqa_linf_step(_hip.ptr(xd), _hip.ptr(gd), _hip.ptr(x0d), _hip.ptr(out), n, 0.01, 0.125,
                                        -1.0, 1.0, 1, None, _hip.stream_for(xd)), "vqa_linf_step")
    assert _same_bits(out, want)


@pytest.mark.parametrize("shape", SHAPES[:3], ids=[str(s) for s in SHAPES[:3]])
def test_linf_fgm_init_project_bitexact(shape):
    ops = _ops()
    x0 = _rand(shape, 10)
    eta = _rand(shape, 11, -0.3, 0.3)
    g = _grad(shape, 12)
    x = torch.clamp(x0 + torch.clamp(eta, -0.125, 0.125), -1, 1)
    assert _same_bits(ops.linf_fgm(x.to(DEV), g.to(DEV), 0.01, -1, 1), o.fgm_update_given_grad(x, g, 0.01, np.inf, -1, 1))
    assert _same_bits(ops.linf_fgm(x.to(DEV), g.to(DEV), 0.01, None, None), o.fgm_update_given_grad(x, g, 0.01, np.inf))
    assert _same_bits(ops.linf_init(x0.to(DEV), eta.to(DEV), 0.125, -1, 1), o.start_point(x0, eta, np.inf, 0.125, -1, 1))
    assert _same_bits(ops.linf_init(x0.to(DEV), None, 0.125, -1, 1), o.start_point(x0, None, np.inf, 0.125, -1, 1))
    far = x0 + eta * 3
    assert _same_bits(ops.linf_project(far.to(DEV), x0.to(DEV), 0.125, -1, 1), o._project(far, x0, np.inf, 0.125, -1, 1))


def test_range_flag():
    ops = _ops()
    x = _rand((2, 3, 8, 8), 13)
    flag = ops.new_flag(DEV)
    ops.linf_init(x.to(DEV), None, 0.1, -1, 1, flag=flag)
    assert int(flag.item()) == 0 and o.range_ok(x, -1, 1)
    for bad in (1.5, -1.0001, float("nan")):
        xb = x.clone()
        xb.view(-1)[17] = bad
        flag = ops.new_flag(DEV)
        ops.linf_init(xb.to(DEV), None, 0.1, -1, 1, flag=flag)
        assert int(flag.item()) == 1 and not o.range_ok(xb, -1, 1)
        flag = ops.new_flag(DEV)
        ops.linf_step(xb.to(DEV), x.to(DEV), x.to(DEV), 0.01, 0.1, -1, 1, flag=flag)
        assert int(flag.item()) == 1


def test_utils_linf_bitexact():
    from vqattack_amd import utils
    t = _grad((6, 3, 5, 4), 14)
    assert _same_bits(utils.clip_eta(t.to(DEV), np.inf, 0.5), o.clip_eta(t.clone(), np.inf, 0.5))
    assert _same_bits(utils.optimize_linear(t.to(DEV), 0.01, np.inf), o.optimize_linear(t.clone(), 0.01, np.inf))


# ----------------------------------------------------------------------------- L2 / L1 (tolerance)
NORM_RTOL = 2e-6


@pytest.mark.parametrize("shape", [(4, 3, 384, 384), (3, 3, 5, 7), (2, 1, 1, 3)], ids=str)
def test_sumsq_and_absmax(shape):
    ops = _ops()
    t = torch.from_numpy(np.random.RandomState(15).standard_normal(shape).astype(np.float32))
    s = _rand(shape, 16)
    got = ops.sumsq_per_sample(t.to(DEV)).cpu().double()
    want = (t.double() ** 2).flatten(1).sum(1)
    assert torch.allclose(got, want, rtol=NORM_RTOL, atol=0)
    got = ops.sumsq_per_sample(t.to(DEV), sub=s.to(DEV)).cpu().double()
    want = ((t - s).double() ** 2).flatten(1).sum(1)
    assert torch.allclose(got, want, rtol=NORM_RTOL, atol=0)
    t.view(shape[0], -1)[0, 0] = 9.5
    t.view(shape[0], -1)[0, -1] = -9.5
    amax, ties = ops.absmax_ties_per_sample(t.to(DEV))
    want_max = t.abs().flatten(1).max(1).values
    assert torch.equal(amax.cpu(), want_max)                      # max is exact
    assert torch.equal(ties.cpu(), (t.abs().flatten(1) == want_max[:, None]).sum(1).float())
    # determinism: bitwise identical on a second run
    assert torch.equal(ops.sumsq_per_sample(t.to(DEV)), ops.sumsq_per_sample(t.to(DEV)))


@pytest.mark.parametrize("shape", [(4, 3, 384, 384), (3, 3, 5, 7)], ids=str)
def test_l2_step_tolerance(shape):
    ops = _ops()
    x0 = _rand(shape, 17)
    x = torch.clamp(x0 + _rand(shape, 18, -0.05, 0.05), -1, 1)
    g = torch.from_numpy(np.random.RandomState(19).standard_normal(shape).astype(np.float32))
    g[0] = 0.0                                                        # avoid_zero_div branch
    eps, eps_iter = 2.0, 0.5
    want = o.pgd_tail_given_grad(x, g, x0, eps_iter, eps, 2, -1, 1)
    mid = ops.l2_fgm(x.to(DEV), g.to(DEV), eps_iter, -1, 1)
    got = ops.l2_project(mid, x0.to(DEV), eps, -1, 1)
    # tolerance: norms agree to 2e-6 relative -> perturbations (|.| <= 2) to ~4e-6 absolute
    assert torch.allclose(got.cpu(), want, rtol=0, atol=5e-6)
    assert torch.allclose(mid.cpu(), o.fgm_update_given_grad(x, g, eps_iter, 2, -1, 1), rtol=0, atol=5e-6)


def test_l1_fgm_and_utils_tolerance():
    ops = _ops()
    from vqattack_amd import utils
    shape = (3, 3, 6, 5)
    x = _rand(shape, 20)
    g = torch.from_numpy(np.random.RandomState(21).standard_normal(shape).astype(np.float32))
    g[1, 0, 0, 0] = 8.0
    g[1, 2, 5, 4] = -8.0          # two-way tie
    want = o.fgm_update_given_grad(x, g, 0.7, 1, -1, 1)
    assert torch.equal(ops.l1_fgm(x.to(DEV), g.to(DEV), 0.7, -1, 1).cpu(), want)   # exact: max/compare/divide by count
    assert torch.equal(utils.optimize_linear(g.to(DEV), 1.5, 1).cpu(), o.optimize_linear(g.clone(), 1.5, 1))
    t = torch.from_numpy(np.random.RandomState(22).standard_normal((6, 3, 5, 4)).astype(np.float32))
    t[1] = 0
    t[3] *= 1e-8
    assert torch.allclose(utils.optimize_linear(t.to(DEV), 0.3, 2).cpu(), o.optimize_linear(t.clone(), 0.3, 2),
                          rtol=5e-6, atol=1e-12)
    td = t.to(DEV)
    r = utils.clip_eta(td, 2, 0.5)
    assert r.data_ptr() == td.data_ptr()                            # in place, like the reference
    assert torch.allclose(r.cpu(), o.clip_eta(t.clone(), 2, 0.5), rtol=5e-6, atol=1e-12)
    with pytest.raises(NotImplementedError):
        utils.clip_eta(td, 1, 0.5)
    with pytest.raises(ValueError):
        utils.clip_eta(td, 3, 0.5)
    with pytest.raises(NotImplementedError):
        utils.optimize_linear(td, 0.5, 3)


# ----------------------------------------------------------------------------- cosine loss
def _cos_ref(a, b, mask=None):
    a = a.clone().requires_grad_(True)
    c = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)(a, b)
    if mask is not None:
        c = c * mask
    loss = torch.sum(-c)
    loss.backward()
    return loss.detach(), a.grad


@pytest.mark.parametrize("rows0,rows1,d", [(13, 617, 768), (3, 17, 16), (25, 40, 1024), (1, 13, 768), (2, 3, 2048),
                                          (2, 5, 260)], ids=str)
def test_neg_cos_rows(rows0, rows1, d):
    ops = _ops()
    r = np.random.RandomState(23)
    a = torch.from_numpy(r.standard_normal((rows0, rows1, d)).astype(np.float32))
    b = torch.from_numpy(r.standard_normal((rows0, rows1, d)).astype(np.float32))
    a[0, 0] = 0.0                                # |a| <= eps branch
    b[0, 1 % rows1] = 0.0
    want_loss, want_grad = _cos_ref(a, b)
    slot = torch.zeros(1, device=DEV)
    ga = ops.neg_cos_rows(a.to(DEV), b.to(DEV), slot, accumulate=False)
    # tolerance: fp32 dot/norm reductions in a different order than ATen -> 1e-5 relative on the loss,
    # 1e-5 of the gradient scale (1/|a|) on the gradient
    assert torch.allclose(slot.cpu()[0], want_loss, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ga.cpu(), want_grad, rtol=1e-4, atol=1e-6)
    # accumulate + negative scale (targeted)
    ops.neg_cos_rows(a.to(DEV), b.to(DEV), slot, accumulate=True, gscale=-1.0, want_grad=False)
    assert abs(float(slot.item())) <= 1e-4 * max(1.0, abs(float(want_loss)))


@pytest.mark.parametrize("d", [768, 1024])
def test_whole_chunk_rows_kernel_equals_the_general_one(d):
    """Rows of NCH * 256 floats of a token map (rows1 > 128) run on the branch-free kernel (csrc/loss.hip,
    neg_cos_rows_full_kernel); the same rows viewed as a map of <= 128 rows per sample -- with the same weights -- run on
    the general one.  Gradients must agree bit for bit (weight-0 rows exactly zero), the loss to fp32 summation order.
    A NaN in a weight-0 row (which only the new kernel reads) must not reach the loss or the gradient."""
    ops = _ops()
    r = np.random.RandomState(31)
    a = torch.from_numpy(r.standard_normal((6, 256, d)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(r.standard_normal((6, 256, d)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((r.uniform(size=(6, 256)) > 0.2).astype(np.uint8))
    w[:, 0] = 2
    w[:, 255] = 0                                           # the last row of every sample is padding
    a[1, 255] = float("nan")
    a[0, 0] = 0.0
    w = w.to(DEV)
    s_full, s_gen = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    g_full = ops.neg_cos_rows(a, b, s_full, accumulate=False, row_weight=w, weight_period=6).clone()
    g_gen = ops.neg_cos_rows(a.view(12, 128, d), b.view(12, 128, d), s_gen, accumulate=False,
                             row_weight=w.view(12, 128), weight_period=12).clone()
    assert torch.equal(g_full.view(-1, d), g_gen.view(-1, d))
    assert not torch.isnan(g_full).any() and float(g_full[1, 255].abs().max()) == 0.0
    assert torch.allclose(s_full, s_gen, rtol=1e-6, atol=0.0)
    # no weights: the whole-chunk kernel for any rows1
    s2 = torch.zeros(1, device=DEV)
    a[1, 255] = 1.0
    g2 = ops.neg_cos_rows(a, b, s2, accumulate=False)
    want_loss, want_grad = _cos_ref(a.cpu(), b.cpu())
    assert torch.allclose(s2.cpu()[0], want_loss, rtol=1e-5, atol=1e-5)
    assert torch.allclose(g2.cpu(), want_grad, rtol=1e-4, atol=1e-6)


def test_neg_cos_rows_strided_views_2d_and_mask():
    ops = _ops()
    r = np.random.RandomState(24)
    full_a = torch.from_numpy(r.standard_normal((6, 20, 64)).astype(np.float32))
    full_b = torch.from_numpy(r.standard_normal((6, 23, 64)).astype(np.float32))
    a, b = full_a[:, :17], full_b[:, :17]                    # the reference's [:, :feat_len, :] truncation
    want_loss, want_grad = _cos_ref(a.contiguous(), b.contiguous())
    slot = torch.zeros(1, device=DEV)
    ga = ops.neg_cos_rows(full_a.to(DEV)[:, :17], full_b.to(DEV)[:, :17], slot, accumulate=False)
    assert ga.shape == a.shape
    assert torch.allclose(slot.cpu()[0], want_loss, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ga.cpu(), want_grad, rtol=1e-4, atol=1e-6)
    a2, b2 = full_a[0], full_b[0, :20]                      # 2-d (rows, D)
    want_loss, want_grad = _cos_ref(a2, b2)
    ga = ops.neg_cos_rows(a2.to(DEV), b2.to(DEV), slot, accumulate=False)
    assert torch.allclose(slot.cpu()[0], want_loss, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ga.cpu(), want_grad, rtol=1e-4, atol=1e-6)
    # row mask with period 2 over the outer axis (layer-major packing of a batch of 2)
    mask = torch.from_numpy((r.uniform(size=(2, 20)) > 0.3).astype(np.uint8))
    mask[0, 0] = 2                                           # a row counted twice (VLMO [CLS])
    a3, b3 = full_a, full_b[:, :20].contiguous()
    m_full = mask.repeat(3, 1).float()                       # outer index o -> mask row o % 2
    want_loss, want_grad = _cos_ref(a3, b3, m_full)
    ga = ops.neg_cos_rows(a3.to(DEV), b3.to(DEV), slot, accumulate=False, row_weight=mask.to(DEV), weight_period=2)
    assert torch.allclose(slot.cpu()[0], want_loss, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ga.cpu(), want_grad, rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------- text side
def test_gather_rows_exact():
    ops = _ops()
    src = torch.from_numpy(np.random.RandomState(25).standard_normal((3, 40, 768)).astype(np.float32))
    for idx in ([1, 3, 4], [39], [0, 0, 5, -1], []):
        got = ops.gather_rows(src.to(DEV), idx)
        assert torch.equal(got.cpu(), src[:, idx])
    small = src[:, :, :6].contiguous()                        # D not a multiple of 4 -> scalar path
    assert torch.equal(ops.gather_rows(small.to(DEV), [2, 7]).cpu(), small[:, [2, 7]])
    with pytest.raises(IndexError):
        ops.gather_rows(src.to(DEV), [40])


@pytest.mark.parametrize("d", [768, 1024, 64])
def test_cand_dir_sim(d):
    ops = _ops()
    from oracle import text_scoring as ts
    r = np.random.RandomState(26)
    vocab, length, nb, k = 500, 12, 3, 4
    f = lambda *s: torch.from_numpy(r.standard_normal(s).astype(np.float32))   # noqa: E731
    word, pos, typ = f(vocab, d) * 0.05, f(64, d) * 0.05, f(2, d) * 0.05
    gamma, beta = 1 + 0.1 * f(d), 0.1 * f(d)
    ids = torch.from_numpy(r.randint(0, vocab, (nb, length)))
    e_ori = ts.bert_embeddings(ids, word, pos, typ, gamma, beta, 1e-12)
    grad = f(nb, k, d)
    cand = torch.tensor([[s, p, kk, v] for s in range(nb) for (p, kk) in ((1, 0), (5, 2), (11, 3))
                         for v in r.randint(0, vocab, 3)], dtype=torch.int32)
    want = ts.candidate_scores(ids, e_ori, grad, cand, word, pos, typ, gamma, beta, 1e-12)
    got = ops.cand_dir_sim(word.to(DEV), pos.to(DEV), typ.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-12,
                           e_ori.to(DEV), grad.to(DEV), cand.to(DEV))
    # tolerance: cosine of fp32 LayerNorm outputs, reduction order differs -> 2e-5 absolute on a value in [-1, 1]
    # (a candidate equal to the original token is never scored: the reference filters it out, adv_attack.py:250-251)
    assert torch.allclose(got.cpu(), want, rtol=0, atol=2e-5)


# ----------------------------------------------------------------------------- BASELINE sizes: properties
@pytest.mark.parametrize("batch", [64, 256])
def test_linf_step_full_size_properties(batch):
    """At BASELINE.json's sizes the CPU oracle is replaced by size-independent properties plus the reference's own
    eager chain evaluated by PyTorch-ROCm on the device (the op chain the authors ran on their GPU)."""
    ops = _ops()
    shape = (batch, 3, 384, 384)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x0 = torch.empty(shape, device=DEV).uniform_(-1, 1, generator=gen)
    eps, eps_iter = 0.125, 0.01
    x = torch.clamp(x0 + torch.empty(shape, device=DEV).uniform_(-eps, eps, generator=gen), -1, 1)
    g = torch.randn(shape, device=DEV, generator=gen)
    g.view(-1)[::1000] = 0
    out = ops.linf_step(x, g, x0, eps_iter, eps, -1, 1)
    # (1) reference chain, eager on the device
    a = torch.clamp(x + eps_iter * torch.sign(g), -1, 1)
    ref = torch.clamp(x0 + torch.clamp(a - x0, -eps, eps), -1, 1)
    assert torch.equal(out, ref)
    del a, ref
    # (2) invariants
    assert float((out - x0).abs().max()) <= np.float32(eps) + 1e-7
    assert float(out.max()) <= 1 and float(out.min()) >= -1
    assert float((out - x).abs().max()) <= np.float32(eps_iter) * 1.0001
    # (3) projection is idempotent on the result; a zero gradient leaves a feasible point unchanged
    assert torch.equal(ops.linf_project(out, x0, eps, -1, 1), out)
    assert torch.equal(ops.linf_step(out, torch.zeros_like(g), x0, eps_iter, eps, -1, 1), out)
    # (4) antisymmetry of the step direction: flipping the gradient flips the move where no clamp is active
    out_neg = ops.linf_step(x, -g, x0, eps_iter, eps, -1, 1)
    inner = ((x - x0).abs() < eps - 2 * eps_iter) & (x.abs() < 1 - 2 * eps_iter)
    # (x + e) - x and x - (x - e) round independently: equal to one ulp of |x| <= 1
    assert torch.allclose((out - x)[inner], -(out_neg - x)[inner], rtol=0, atol=2e-7)


# ----------------------------------------------------------------------------- fused MLM cross entropy
@pytest.mark.parametrize("rows,k", [(12, 1), (7, 3), (2560, 1), (40, 5)], ids=str)
def test_mlm_cross_entropy(rows, k):
    import torch.nn.functional as F
    ops = _ops()
    r = np.random.RandomState(27)
    v = 30522
    logits = torch.from_numpy((r.standard_normal((rows, v)) * 3).astype(np.float32))
    labels = torch.from_numpy(r.randint(0, v, (k, rows)))
    labels[:, ::3] = -100
    if k > 1:
        labels[1, :] = -100 if rows < 10 else labels[1, :]      # a fully ignored label set on the small case
        labels[1, 1] = 5
    a = logits.clone().requires_grad_(True)
    want = sum(F.cross_entropy(a, labels[i], ignore_index=-100) for i in range(k))
    want.backward()
    slot = torch.zeros(1, device=DEV)
    g = ops.mlm_cross_entropy(logits.to(DEV), labels.to(DEV), slot, accumulate=False)
    # tolerance: fp32 log-sum-exp over 30522 terms in a different order than ATen: 1e-5 relative on the loss,
    # 1e-6 absolute + 1e-4 relative on gradient entries (each <= 1/n_valid)
    assert torch.allclose(slot.cpu()[0], want.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(g.cpu(), a.grad, rtol=1e-4, atol=1e-7)
    # loss only + scaling
    ops.mlm_cross_entropy(logits.to(DEV), labels.to(DEV), slot, accumulate=True, gscale=-1.0, want_grad=False)
    assert abs(float(slot.item())) <= 1e-4 * max(1.0, abs(float(want.detach() if torch.is_tensor(want) else want)))
    # 3-d view of the logits (B, L, V) as the adapters return them
    if rows % 4 == 0:
        g3 = ops.mlm_cross_entropy(logits.to(DEV).reshape(4, rows // 4, v), labels.to(DEV), slot, accumulate=False)
        assert g3.shape == (4, rows // 4, v) and torch.equal(g3.reshape(rows, v), g)


# ----------------------------------------------------------------------------- masked-token embedding substitution
@pytest.mark.parametrize("d", [768, 1024, 64])
def test_embed_tokens(d):
    ops = _ops()
    from oracle import text_scoring as ts
    r = np.random.RandomState(28)
    vocab, length, nb = 700, 10, 5
    f = lambda *s: torch.from_numpy(r.standard_normal(s).astype(np.float32))   # noqa: E731
    word, pos, typ = f(vocab, d) * 0.05, f(64, d) * 0.05, f(2, d) * 0.05
    gamma, beta = 1 + 0.1 * f(d), 0.1 * f(d)
    tables = dict(word=word.to(DEV), pos=pos.to(DEV), type_emb=typ.to(DEV), gamma=gamma.to(DEV), beta=beta.to(DEV),
                  ln_eps=1e-12)
    ids = torch.from_numpy(r.randint(0, vocab, (nb, length)))
    want = ts.bert_embeddings(ids, word, pos, typ, gamma, beta, 1e-12)
    got = ops.embed_tokens(tables, ids.to(DEV))
    # tolerance: fp32 LayerNorm with a different reduction order than ATen: 2e-6 absolute on O(1) values
    assert torch.allclose(got.cpu(), want, rtol=0, atol=2e-6)
    # substitute two words and rewrite only their rows
    new_ids = ids.clone()
    new_ids[1, 3], new_ids[4, 7] = 5, 6
    before = got.clone()
    ops.embed_tokens(tables, new_ids.to(DEV), out=got, rows=[(1, 3), (4, 7)])
    want2 = ts.bert_embeddings(new_ids, word, pos, typ, gamma, beta, 1e-12)
    assert torch.allclose(got.cpu(), want2, rtol=0, atol=2e-6)
    untouched = torch.ones(nb, length, dtype=torch.bool)
    untouched[1, 3] = untouched[4, 7] = False
    assert torch.equal(got.cpu()[untouched], before.cpu()[untouched])


# ----------------------------------------------------------------------------- empty inputs
def test_empty_inputs_are_no_ops():
    ops = _ops()
    from vqattack_amd import utils
    e = torch.empty(0, 3, 8, 8, device=DEV)
    assert ops.linf_step(e, e, e, 0.01, 0.125, -1, 1).shape == e.shape
    assert ops.linf_init(e, None, 0.125, -1, 1).shape == e.shape
    assert utils.clip_eta(e, np.inf, 0.1).shape == e.shape
    assert utils.optimize_linear(e, 0.1, np.inf).shape == e.shape
    assert ops.sumsq_per_sample(e).shape == (0,)
    slot = torch.ones(1, device=DEV)
    ga = ops.neg_cos_rows(torch.empty(0, 5, 16, device=DEV), torch.empty(0, 5, 16, device=DEV), slot, accumulate=False)
    assert ga.shape == (0, 5, 16) and float(slot) == 0.0
    assert ops.gather_rows(torch.empty(0, 4, 16, device=DEV), [1, 2]).shape == (0, 2, 16)


@pytest.mark.parametrize("v", [1000, 7, 33334, 30523], ids=str)
def test_mlm_cross_entropy_other_vocabularies(v):
    """Vocabulary sizes off the register-resident fast path (tiny, larger than one workgroup's registers, odd)."""
    import torch.nn.functional as F
    ops = _ops()
    r = np.random.RandomState(29)
    rows = 9
    logits = torch.from_numpy((r.standard_normal((rows, v)) * 2).astype(np.float32))
    labels = torch.from_numpy(r.randint(0, v, (2, rows)))
    labels[0, ::2] = -100
    a = logits.clone().requires_grad_(True)
    want = sum(F.cross_entropy(a, labels[i], ignore_index=-100) for i in range(2))
    want.backward()
    slot = torch.zeros(1, device=DEV)
    g = ops.mlm_cross_entropy(logits.to(DEV), labels.to(DEV), slot, accumulate=False)
    assert torch.allclose(slot.cpu()[0], want.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(g.cpu(), a.grad, rtol=1e-4, atol=1e-7)


def test_neg_cos_rows_multi_equals_per_layer_launches():
    ops = _ops()
    r = np.random.RandomState(30)
    layers = 5
    a = [torch.from_numpy(r.standard_normal((3, 17, 64)).astype(np.float32)).to(DEV) for _ in range(layers)]
    b = [torch.from_numpy(r.standard_normal((3, 17, 64)).astype(np.float32)).to(DEV) for _ in range(layers)]
    w = torch.from_numpy((r.uniform(size=(3, 17)) > 0.2).astype(np.uint8)).to(DEV)
    w[:, 0] = 2
    one, many = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    g_one = [ops.neg_cos_rows(a[i], b[i], one, accumulate=i > 0, gscale=0.5, row_weight=w, weight_period=3)
             for i in range(layers)]
    g_many = ops.neg_cos_rows_multi(a, b, many, accumulate=False, gscale=0.5, row_weight=w, weight_period=3)
    assert len(g_many) == layers
    for x, y in zip(g_one, g_many):
        assert torch.equal(x, y)                                   # same arithmetic per row
    assert torch.allclose(one, many, rtol=1e-6, atol=1e-5)        # partial sums are grouped differently
    # loss only, and the non-uniform fallback (one layer is a strided view with different strides)
    ops.neg_cos_rows_multi(a, b, many, accumulate=True, gscale=-0.5, want_grad=False, row_weight=w, weight_period=3)
    assert abs(float(many)) <= 1e-4
    big = torch.zeros(3, 20, 64, device=DEV)
    big[:, :17] = a[2]
    a_mixed = list(a)
    a_mixed[2] = big[:, :17]
    g_fb = ops.neg_cos_rows_multi(a_mixed, b, many, accumulate=False, gscale=0.5, row_weight=w, weight_period=3)
    for x, y in zip(g_one, g_fb):
        assert torch.equal(x, y)


def test_in_kernel_loss_fold_is_reproducible_and_equals_the_two_launch_form():
    """The last-arriving workgroup folds the partials in index order: same bits run to run, and the same value as
    partials + vqa_sum_partials."""
    from vqattack_amd import _hip
    ops = _ops()
    r = np.random.RandomState(91)
    a = torch.from_numpy(r.standard_normal((64, 617, 768)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(r.standard_normal((64, 617, 768)).astype(np.float32)).to(DEV)
    slot = torch.zeros(1, device=DEV)
    vals = []
    for _ in range(5):
        ops.neg_cos_rows(a, b, slot, accumulate=False)
        vals.append(slot.clone())
    assert all(torch.equal(v, vals[0]) for v in vals)
    # partials-only launch + explicit fold through the C ABI
    part = torch.zeros(_hip.lib().vqa_neg_cos_partials(), device=DEV)
    ga = torch.empty_like(a)
    lib = _hip.lib()
    _hip.check(lib.vqa_neg_cos_rows(_hip.ptr(a), _hip.ptr(b), _hip.ptr(ga), _hip.ptr(part), None, 1, 64, 617, 768,
                                    617 * 768, 768, 617 * 768, 768, 617 * 768, 768, 1.0, 1e-6, None, 0,
                                    _hip.stream_for(a)), "vqa_neg_cos_rows")
    two = torch.zeros(1, device=DEV)
    _hip.check(lib.vqa_sum_partials(_hip.ptr(part), part.numel(), _hip.ptr(two), 0, 1.0, _hip.stream_for(a)), "sum")
    assert torch.allclose(two, vals[0], rtol=1e-6)
    # accumulate on top of an existing slot value
    slot.fill_(3.0)
    ops.neg_cos_rows(a, b, slot, accumulate=True, want_grad=False)
    assert torch.allclose(slot, vals[0] + 3.0, rtol=1e-6)


def test_mlm_cross_entropy_flags_out_of_range_labels():
    """A label that is neither ignore_index nor in [0, V): NaN loss + VQA_FLAG_BAD_LABEL (torch device-asserts)."""
    from vqattack_amd import _hip
    ops = _ops()
    logits = torch.randn(6, 30522, device=DEV)
    labels = torch.tensor([[5, -100, 30522, 7, -100, 9]], device=DEV)
    slot, flag = torch.zeros(1, device=DEV), ops.new_flag(DEV)
    g = ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, flag=flag)
    assert torch.isnan(slot).all() and int(flag.item()) & _hip.VQA_FLAG_BAD_LABEL
    assert torch.isfinite(g[[0, 1, 3, 4, 5]]).all()
    labels[0, 2] = -7
    slot.zero_(), flag.zero_()
    ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, flag=flag, want_grad=False)
    assert torch.isnan(slot).all() and int(flag.item()) & _hip.VQA_FLAG_BAD_LABEL
    labels[0, 2] = -100
    slot.zero_(), flag.zero_()
    ops.mlm_cross_entropy(logits, labels, slot, accumulate=False, flag=flag)
    assert torch.isfinite(slot).all() and int(flag.item()) == 0


def test_mlm_cross_entropy_dead_rows_are_not_stored_with_a_workspace():
    """External dense ``model_fn`` (logits (B, L, V), labels ignore_index except at the [MASK]-ed answer pieces,
    adv_attack.py:433-558) with the attack's workspace: the gradient buffer is zero-filled once, a byte per row remembers
    which rows hold a live gradient, dead rows are never stored again.  The result must equal the workspace-free launch
    on every call -- also when the label pattern CHANGES between calls (rows that were live turn dead and have to be
    zeroed exactly once) and when the batch shrinks (prefix view of the same buffer)."""
    ops = _ops()
    r = np.random.RandomState(11)
    b, l, v = 6, 10, 30522
    logits = torch.from_numpy(r.standard_normal((b, l, v)).astype(np.float32)).to(DEV)

    def labels_with(live):
        lab = torch.full((1, b * l), -100, dtype=torch.long)
        for row in live:
            lab[0, row] = int(r.randint(0, v))
        return lab.to(DEV)

    ws = ops.Workspace()
    slot, ref_slot = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    patterns = [[3, 14, 27, 41, 55], [3, 14, 27, 41, 55], [4, 14, 30], [], [0, 59], [4, 14, 30]]
    for step, live in enumerate(patterns):
        lab = labels_with(live)
        want = ops.mlm_cross_entropy(logits, lab, ref_slot, accumulate=False, rows_per_sample=l)           # fresh buffer
        got = ops.mlm_cross_entropy(logits, lab, slot, accumulate=False, rows_per_sample=l, ws=ws)
        assert torch.equal(got, want), (step, live)
        assert torch.equal(slot, ref_slot) or (torch.isnan(slot).all() and torch.isnan(ref_slot).all())
        state = ws.get(("ce_row_state", v), (b * l,), torch.uint8, logits.device, zero=True)
        assert sorted(torch.nonzero(state).flatten().tolist()) == sorted(live)
        dead = [i for i in range(b * l) if i not in live]
        assert float(got.reshape(b * l, v)[dead].abs().max()) == 0.0
    # a shrunken batch (attack_mixed: finished samples leave the batch) takes a prefix view of the same buffers
    lab = labels_with([2, 17])[:, :3 * l].contiguous()
    want = ops.mlm_cross_entropy(logits[:3], lab, ref_slot, accumulate=False, rows_per_sample=l)
    got = ops.mlm_cross_entropy(logits[:3], lab, slot, accumulate=False, rows_per_sample=l, ws=ws)
    assert torch.equal(got, want) and got.data_ptr() == ws.get(("ce_grad", v), (1,), torch.float32, logits.device).data_ptr()


def test_optimize_linear_self_checks_are_reported_not_dropped():
    """The reference asserts inside optimize_linear that its result has unit norm (utils.py:101-104 L1, :110-116 L2).
    The only inputs that trip it -- an all-zero or NaN L1 gradient, a non-finite L2 norm -- raise AssertionError here
    too (flag bit VQA_FLAG_DEGENERATE, one host read), standalone and inside the FGM / PGD operators."""
    from vqattack_amd import _hip, attacks, utils
    ops = _ops()
    g = torch.randn(3, 3, 8, 8, device=DEV)
    for norm in (1, 2):
        utils.optimize_linear(g, 0.5, norm)                          # healthy gradients: no assert
    bad = g.clone()
    bad[1] = 0.0
    utils.optimize_linear(bad, 0.5, 2)                               # L2 of a zero gradient is fine (avoid_zero_div)
    with pytest.raises(AssertionError):
        utils.optimize_linear(bad, 0.5, 1)                           # L1: every entry ties at 0 with sign 0
    nan = g.clone()
    nan[2, 0, 0, 0] = float("nan")
    inf = g.clone()
    inf[0, 1, 2, 3] = float("inf")
    for t, norm in ((nan, 1), (nan, 2), (inf, 2)):
        with pytest.raises(AssertionError):
            utils.optimize_linear(t, 0.5, norm)
    utils.optimize_linear(inf, 0.5, 1)                               # one infinite entry: sign * 1 / 1, unit L1 norm
    # the fused FGM updates report the same bit into the attack's flag word
    x = torch.zeros_like(g)
    flag = ops.new_flag(DEV)
    ops.l2_fgm(x, g, 0.1, -1, 1, flag=flag)
    ops.l1_fgm(x, g, 0.1, -1, 1, flag=flag)
    assert int(flag.item()) == 0
    ops.l1_fgm(x, bad, 0.1, -1, 1, flag=flag)
    assert int(flag.item()) == _hip.VQA_FLAG_DEGENERATE
    flag.zero_()
    ops.l2_fgm(x, inf, 0.1, -1, 1, flag=flag, check_range=False)
    assert int(flag.item()) == _hip.VQA_FLAG_DEGENERATE

    # through the operator API: a model whose gradient is non-finite for one sample
    def model_fn(img):
        f = (img * torch.tensor([1.0, float("inf"), 1.0], device=DEV).view(3, 1, 1, 1)).reshape(3, 1, -1)
        return [f, f * 2.0]

    y = [torch.ones(3, 1, 3 * 8 * 8, device=DEV)]
    with pytest.raises(AssertionError):
        attacks.projected_gradient_descent(model_fn, x + 0.1, 0.5, 0.1, 2, 2, clip_min=-1, clip_max=1, y=y + y,
                                           ori_x=x + 0.1, time=1, ls=1, sanity_checks=False, flavor="albef")


@pytest.mark.parametrize("k", [1, 3])
def test_mlm_cross_entropy_per_sample_normalisation(k):
    """rows_per_sample = L: the loss is the SUM over samples of the batch-1 reference loss (each sample's label sets
    divided by that sample's own valid counts), gradients per sample equal the batch-1 gradients; an all-ignored label
    set of one sample contributes nothing."""
    import torch.nn.functional as F
    ops = _ops()
    b, l, v = 5, 9, 30522
    r = np.random.RandomState(7)
    logits = torch.from_numpy(r.standard_normal((b, l, v)).astype(np.float32))
    labels = torch.full((b, k, l), -100, dtype=torch.long)
    for s in range(b):
        for j in range(k):
            if j > 0 and s % 2 == 0:
                continue                       # ragged K: this sample has fewer label sets
            n_valid = 1 + (s + j) % 3
            pos = r.choice(np.arange(1, l), n_valid, replace=False)
            labels[s, j, pos] = torch.from_numpy(r.randint(0, v, n_valid))
    lg = logits.clone().requires_grad_(True)
    want = 0.0
    for s in range(b):
        for j in range(k):
            if (labels[s, j] != -100).any():
                want = want + F.cross_entropy(lg[s], labels[s, j], ignore_index=-100)
    want.backward()
    sets = labels.permute(1, 0, 2).reshape(k, -1).to(DEV)
    slot = torch.zeros(1, device=DEV)
    g = ops.mlm_cross_entropy(logits.to(DEV), sets, slot, accumulate=False, rows_per_sample=l)
    assert torch.allclose(slot.cpu()[0], want.detach(), rtol=1e-4)
    assert torch.allclose(g.cpu(), lg.grad, rtol=1e-4, atol=1e-7)


def test_greedy_accept_on_device_equals_host_loop():
    """vqa_greedy_accept (one wave per sample, device-resident ids) vs the host acceptance loop with the same
    bag-of-embeddings similarity, random candidates incl. ties, repeated positions, padding and empty samples."""
    from vqattack_amd.attack import text_update
    ops = _ops()
    r = np.random.RandomState(3)
    b, l, v, e = 37, 23, 500, 64
    table = r.standard_normal((v, e)).astype(np.float32)
    ori = r.randint(1, v, size=(b, l)).astype(np.int64)
    for s in range(b):
        ori[s, 4 + s % (l - 4):] = 0
    cur = ori.copy()
    cur[:, 2] = r.randint(1, v, size=b)                     # already one substitution behind
    proposals = []
    for s in range(b):
        per = [] if s % 9 == 0 else [(int(p), [int(x) for x in r.randint(1, v, 5)])
                                      for p in r.choice(np.arange(1, 4 + s % (l - 4)), min(3, 3 + s % (l - 4)), replace=False)]
        proposals.append(per)
    plan = text_update.CandidatePlan(proposals, DEV)
    scores = r.standard_normal(len(plan)).astype(np.float32)
    scores[::7] = scores[1::7][:len(scores[::7])] if len(scores) > 8 else scores[::7]     # ties
    sim = text_update.BagOfEmbeddingsSimilarity(table=table)
    for thr in (0.95, 0.5, -1.0):
        want_ids, want_ops = text_update.greedy_accept(plan.rows, scores, ori, cur, sim, thr)
        got = torch.from_numpy(cur).to(DEV)
        prev = got.clone()
        new_id, rank = ops.greedy_accept(plan.device_rows, torch.from_numpy(scores).to(DEV),
                                         torch.from_numpy(ori).to(DEV), got, sim.device_table(DEV), thr)
        assert np.array_equal(got.cpu().numpy(), want_ids), thr
        assert text_update.substitution_lists(prev, new_id, rank) == want_ops
    assert any(len(o) for o in want_ops)


def test_a_bad_mlm_label_fails_the_step_also_without_sanity_checks():
    """``F.cross_entropy`` with a target outside the vocabulary never returns in the reference, whatever ``sanity_checks``
    says (fast_gradient_method.py:133-139: IndexError on the host, a device assert on a GPU); here the CE kernel sets
    VQA_FLAG_BAD_LABEL and the norm-1 / norm-2 operators -- which read the flag word anyway for optimize_linear's own
    assert -- raise on it instead of stepping along a NaN gradient."""
    from vqattack_amd import attacks
    g = torch.Generator(device=DEV).manual_seed(0)
    w = torch.randn(3 * 8 * 8, 30522, device=DEV, generator=g) * 0.01
    x = torch.empty(2, 3, 8, 8, device=DEV).uniform_(-0.5, 0.5, generator=g)

    def model_fn(img):
        return [(img.reshape(2, 1, -1) @ w)]                       # logits (B, L = 1, V)

    good = torch.tensor([[5], [77]], device=DEV)
    bad = torch.tensor([[5], [30522]], device=DEV)
    for norm in (1, 2):
        adv, loss = attacks.fast_gradient_method(model_fn, x, 0.1, norm, x, clip_min=-1, clip_max=1, y=[good], ls=0,
                                                 sanity_checks=False, flavor="albef")
        assert torch.isfinite(adv).all()
        with pytest.raises(AssertionError, match="MLM label"):
            attacks.fast_gradient_method(model_fn, x, 0.1, norm, x, clip_min=-1, clip_max=1, y=[bad], ls=0,
                                         sanity_checks=False, flavor="albef")
