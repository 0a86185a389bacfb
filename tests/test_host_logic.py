"""CPU-side checks of the shipped host code: argument validation happens before any device work, CPU tensors are
refused loudly (no fallback), and the drop-in packages expose the reference's module paths."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import vqattack_amd
from vqattack_amd._hip import HipExtensionError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X = torch.zeros(1, 3, 4, 4)


def _fn(t):
    return [t, t]


@pytest.mark.parametrize("flavor", ["albef", "vlmo"])
def test_validation_precedes_device_checks(flavor):
    from vqattack_amd import dropin
    pgd = dropin.load(flavor).projected_gradient_descent.projected_gradient_descent
    fgm = dropin.load(flavor).fast_gradient_method.fast_gradient_method
    with pytest.raises(NotImplementedError):
        pgd(_fn, X, 0.1, 0.01, 1, 1, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, 0.01, 1, 7, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, -1, 0.01, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, -1, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(AssertionError):
        pgd(_fn, X, 0.1, 0.5, 1, np.inf, ori_x=X, ls=1)
    with pytest.raises(ValueError):
        pgd(_fn, X, 0.1, 0.01, 1, np.inf, clip_min=2, clip_max=1, ori_x=X, ls=1)
    assert pgd(_fn, X, 0, 0.01, 1, np.inf, ori_x=X, ls=1) is X
    assert pgd(_fn, X, 0.1, 0, 1, np.inf, ori_x=X, ls=1) is X
    with pytest.raises(ValueError):
        fgm(_fn, X, 0.1, 5, X, ls=1)
    with pytest.raises(ValueError):
        fgm(_fn, X, -0.1, np.inf, X, ls=1)
    assert fgm(_fn, X, 0, np.inf, X, ls=1) is X


def test_cpu_tensors_are_refused_not_emulated():
    with pytest.raises(HipExtensionError, match="no CPU fallback"):
        vqattack_amd.projected_gradient_descent(_fn, X, 0.1, 0.01, 1, np.inf, ori_x=X, ls=1, y=[X, X])
    with pytest.raises(HipExtensionError):
        vqattack_amd.fast_gradient_method(_fn, X, 0.1, np.inf, X, ls=1, y=[X, X])
    with pytest.raises(HipExtensionError):
        vqattack_amd.clip_eta(X, np.inf, 0.1)
    with pytest.raises(HipExtensionError):
        vqattack_amd.optimize_linear(X, 0.1, 2)
    with pytest.raises(NotImplementedError):        # reference behaviour, raised before the device check
        vqattack_amd.clip_eta(X, 1, 0.1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vqattack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, f)


@pytest.mark.parametrize("flavor", ["albef", "vlmo"])
def test_dropin_module_paths(flavor):
    root = os.path.join(ROOT, "vqattack_amd", "dropin", flavor)
    for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
        del sys.modules[name]
    sys.path.insert(0, root)
    try:
        pgd = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent")
        pgd_vl = importlib.import_module("cleverhans.torch.attacks.projected_gradient_descent_vl")
        fgm = importlib.import_module("cleverhans.torch.attacks.fast_gradient_method")
        fgm_vl = importlib.import_module("cleverhans.torch.attacks.fast_gradient_method_vl")
        utils = importlib.import_module("cleverhans.torch.utils")
        assert pgd.projected_gradient_descent.keywords == {"flavor": flavor}
        assert pgd_vl.projected_gradient_descent.keywords == {"flavor": flavor}
        assert fgm.fast_gradient_method.keywords == {"flavor": flavor}
        assert fgm_vl.fast_gradient_method.keywords == {"flavor": flavor}
        assert callable(utils.clip_eta) and callable(utils.optimize_linear)
    finally:
        sys.path.remove(root)
        for name in [m for m in sys.modules if m == "cleverhans" or m.startswith("cleverhans.")]:
            del sys.modules[name]


def test_attention_bias_layout_decision():
    """Which bias tensors the attention kernels may read in place (rows readable to ceil32(Sk), aligned strides) and
    which must be copied into padded rows first -- the rule of `attention._bias_layout`, on host tensors."""
    import torch
    from vqattack_amd import attention

    s = 587
    slab = torch.zeros(1, 12, s, 608)                       # what FrozenVlmo.attention_bias builds: rows padded to 32
    view, ok = attention._bias_layout(slab[..., :s].expand(64, -1, -1, -1), s)
    assert ok and view.shape == (1, 12, s, s) and view.data_ptr() == slab.data_ptr()
    _, ok = attention._bias_layout(torch.zeros(1, 12, s, s), s)          # dense rows of 587 floats: no room, odd pitch
    assert not ok
    _, ok = attention._bias_layout(torch.zeros(2, 3, 64, 64), 64)        # Sk a multiple of 32: dense rows are fine
    assert ok
    tight = torch.zeros(1, 1, 4, 96)[..., :70]                           # pitch 96 = ceil32(70): exactly enough room
    assert attention._bias_layout(tight, 70)[1]
    short = torch.zeros(1, 1, 4, 80)[..., :70]                           # last row ends 16 floats early
    assert not attention._bias_layout(short, 70)[1]
    pad_mask = torch.zeros(8, 1, 1, 40).expand(8, 12, 40, 40)            # key padding broadcast over heads and rows
    view, ok = attention._bias_layout(pad_mask, 40)
    assert view.shape == (8, 1, 1, 40) and not ok                        # 40-float rows: copied, but only 8 of them
    with pytest.raises(ValueError):
        attention._bias_layout(torch.zeros(1, 1, 4, 33), 40)


def _tiny(flavor):
    if flavor == "vlmo":
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
        cfg = vlmo_tiny()
        return FrozenVlmo(cfg, seed=2), VlmoAttackAdapters, cfg
    from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_tiny
    cfg = albef_tiny(mlm_probability=0.0)
    return FrozenAlbef(cfg, seed=2), AlbefAttackAdapters, cfg


def test_live_label_rows():
    """mlm_task.live_label_rows: live positions first (ascending), padded with dead positions of the same sample."""
    from vqattack_amd.attack import mlm_task
    lab = torch.full((3, 2, 8), -100, dtype=torch.long)
    lab[0, 0, 5] = 11
    lab[1, 0, 2], lab[1, 1, 2], lab[1, 0, 6], lab[1, 1, 3] = 21, 22, 23, 24          # live: 2, 3, 6
    rows, compact = mlm_task.live_label_rows(lab)
    assert rows.shape == (3, 3) and compact.shape == (3, 2, 3)
    assert rows[0].tolist()[0] == 5 and rows[1].tolist() == [2, 3, 6]
    assert compact[1].tolist() == [[21, -100, 23], [22, 24, -100]]
    assert compact[0].tolist() == [[11, -100, -100], [-100, -100, -100]] and (compact[2] == -100).all()
    assert len(set(rows[0].tolist())) == 3 and all(lab[0, :, p].eq(-100).all() for p in rows[0].tolist()[1:])
    rows2, compact2 = mlm_task.live_label_rows(lab[:, 0])
    assert rows2.shape == (3, 2) and compact2[1].tolist() == [21, 23] and compact2[0].tolist()[0] == 11
    rows3, compact3 = mlm_task.live_label_rows(torch.full((2, 8), -100, dtype=torch.long))   # nothing live: width 1
    assert rows3.shape == (2, 1) and (compact3 == -100).all()


@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_live_rows_and_mixed_closures_equal_the_dense_closures(flavor):
    """The live-rows form of ``pgd_mlm_attack`` returns the dense closure's logits at the live label rows, and
    ``pgd_attack_mixed`` returns, per sample, what ``pgd_attack_vl`` (feature step) or ``pgd_mlm_attack`` (MLM step on the
    paraphrase) return for that sample -- on host tensors (the frozen white boxes are plain PyTorch modules)."""
    from vqattack_amd.attack import mlm_task
    model, adapters_cls, cfg = _tiny(flavor)
    a = adapters_cls(model)
    g = torch.Generator().manual_seed(3)
    length = cfg.max_text_len if flavor == "vlmo" else 8
    ids = torch.zeros(3, length, dtype=torch.long)
    para = torch.zeros(3, length, dtype=torch.long)
    for s, (n, m) in enumerate([(3, 5), (5, 4), (2, 6)]):
        ids[s, 0], ids[s, 1 + n] = 101, 102
        ids[s, 1:1 + n] = torch.randint(1000, 30522, (n,), generator=g)
        para[s, 0], para[s, 1 + m] = 101, 102
        para[s, 1:1 + m] = torch.randint(1000, 30522, (m,), generator=g)
    masks, pmasks = (ids != 0).long(), (para != 0).long()
    labels = torch.full((3, 2, length), -100, dtype=torch.long)
    labels[0, 0, 2], labels[1, 0, 1], labels[1, 1, 3], labels[2, 0, 4] = 5, 6, 7, 8
    image = torch.empty(3, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    rows, compact = mlm_task.live_label_rows(labels)
    a.set_text(ids, masks, text_ids_mlm=para, text_mask_mlm=pmasks)
    pinned = a._tlen
    dense = a.pgd_mlm_attack(image)[0]
    a.set_mlm_rows(rows)
    live = a.pgd_mlm_attack(image)[0]
    assert live.shape == (3, rows.shape[1], cfg.vocab)
    want = torch.gather(dense, 1, rows.unsqueeze(-1).expand(-1, -1, cfg.vocab))
    assert torch.allclose(live, want, rtol=1e-5, atol=1e-5)
    assert compact.shape == (3, 2, rows.shape[1])
    # mixed step: samples 0 and 2 take a feature step (question text), sample 1 an MLM step (paraphrase text)
    emb_q, emb_p = model.text_embeddings(ids), model.text_embeddings(para)
    feats_all = a.pgd_attack_vl([image, emb_q])
    sel = torch.tensor([1])
    ids_t, masks_t, emb_t = ids.clone(), masks.clone(), emb_q.clone()
    ids_t[sel], masks_t[sel], emb_t[sel] = para[sel], pmasks[sel], emb_p[sel]
    a.set_text(ids_t, masks_t, text_len=pinned)
    a.set_mlm_samples(sel)
    out, logits = a.pgd_attack_mixed([image, emb_t])
    assert logits.shape == (1, rows.shape[1], cfg.vocab)
    assert torch.allclose(logits[0], live[1], rtol=1e-4, atol=1e-4)
    lf_mixed, lf_all = (out[2], feats_all[2]) if flavor == "vlmo" else (out[1], feats_all[1])
    for got, ref in zip(lf_mixed.layers, lf_all.layers):
        assert torch.allclose(got[[0, 2]], ref[[0, 2]], rtol=1e-4, atol=1e-4)
    assert int(lf_mixed.row_weight[1].sum()) == 0 and int(lf_mixed.row_weight[0].sum()) > 0
    if flavor == "albef":
        assert int(out[0].row_weight[1].sum()) == 0
    # everybody at an MLM step: no feature list at all
    a.set_text(para, pmasks, text_len=pinned)
    a.set_mlm_samples(torch.arange(3))
    out, logits = a.pgd_attack_mixed([image, emb_p])
    assert out is None and torch.allclose(logits, live, rtol=1e-4, atol=1e-4)
    # nobody: the plain probe closure; snapshots restore a text batch without rebuilding it
    a.set_text(ids, masks, text_len=pinned)
    state = a.save_text()
    a.set_mlm_samples(None)
    out, logits = a.pgd_attack_mixed([image, emb_q])
    assert logits is None and len(out) == len(feats_all)
    a.set_text(para, pmasks, text_len=pinned)
    a.load_text(state)
    assert torch.equal(a.batch["text_ids"], ids[:, :pinned])


def test_step_kinds_follow_the_reference_schedule():
    """Per-sample step sequences of attack_mixed: feature samples F^blocks with probes in between; dual samples
    (N, M)^(block // 2); both take budget + words white-box gradient steps."""
    from vqattack_amd.attack.runner import BatchedVQAttack
    from vqattack_amd.attack.schedule import iter_schedule
    for words in range(0, 7):
        blocks = iter_schedule(words, 40) or [40]
        feat = BatchedVQAttack._step_kinds(words, 40, False)
        dual = BatchedVQAttack._step_kinds(words, 40, True)
        assert len(feat) == len(dual) == 40 + words
        assert sum(p for _, p in feat) == sum(p for _, p in dual) == len(blocks) - 1
        assert all(k == "F" for k, _ in feat)
        at = 0
        for j, blen in enumerate(blocks):
            assert [k for k, _ in dual[at:at + blen]] == ["N", "M"] * (blen // 2)
            at += blen
            if j < len(blocks) - 1:
                assert dual[at] == ("F", True) and feat[at] == ("F", True)
                at += 1
    assert [k for k, _ in BatchedVQAttack._step_kinds(0, 7, True)] == ["N", "M"] * 3      # int(7 / 2) dual iterations


def test_mixed_update_runs_and_paraphrase_following(monkeypatch):
    """Host logic of ``attack_mixed`` that needs no device: the fused update is launched once per run of consecutive
    samples of one kind (``N`` = no projection -> vqa_linf_fgm; everything else -> vqa_linf_step), and a dual-loss
    sample's paraphrase takes over the question's substitutions like ``update_mlm_text`` (adv_attack.py:334-351)."""
    from vqattack_amd import ops
    from vqattack_amd.attack import mlm_task
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    calls = []
    monkeypatch.setattr(ops, "linf_fgm", lambda x, g, *a, **kw: calls.append(("fgm", x.shape[0], int(x[0, 0]))))
    monkeypatch.setattr(ops, "linf_step", lambda x, g, x0, *a, **kw: calls.append(("step", x.shape[0], int(x[0, 0]))))
    attack = BatchedVQAttack.__new__(BatchedVQAttack)
    attack.cfg = AttackConfig()
    cur = torch.arange(7, dtype=torch.float32).reshape(7, 1).repeat(1, 2)      # sample s is filled with s
    attack._update_runs(cur, cur.clone(), cur.clone(), ["F", "F", "N", "N", "M", "N", "F"][:6])
    assert calls == [("step", 2, 0), ("fgm", 2, 2), ("step", 1, 4), ("fgm", 1, 5)]   # the 7th sample is not active
    calls.clear()
    attack._update_runs(cur, cur.clone(), cur.clone(), ["M"] * 5)
    assert calls == [("step", 5, 0)]
    # update_mlm_text: one-piece words equal to a substituted question word follow it, every occurrence
    task = mlm_task.build_mlm_task([(7001,)], [[(7001,)]], [True], [(2054,), (3609,), (7001,), (2054,)], [], "vlmo",
                                   max_len=12)
    tasks = [None, mlm_task.MlmTask(**vars(task))]
    tasks[1].words_mlm = list(task.words_mlm)
    ids = torch.zeros(2, 12, dtype=torch.long)
    msk = torch.zeros(2, 12, dtype=torch.long)
    BatchedVQAttack._write_mlm_row(ids, msk, 1, tasks[1])
    before = ids[1].tolist()
    assert before[:6] == [101, 2054, 3609, 103, 2054, 102] and msk[1].tolist()[:7] == [1] * 6 + [0]
    changed = attack._follow_substitutions(tasks, [1], ids, msk, [[], [(1, 2054, 9999)]])
    assert changed and ids[1].tolist()[:6] == [101, 9999, 3609, 103, 9999, 102]
    assert not attack._follow_substitutions(tasks, [1], ids, msk, [[], []])
    assert task.words_mlm[0] == (2054,)                                       # the caller's task is untouched


def test_key_hole_bias_densifies_to_the_per_sample_padding_mask():
    """``attention.KeyHoleBias`` (one relative-position slab + per-sample masked key range, what a ragged batch hands to
    the GPU attention kernel) must describe exactly the per-sample additive mask the host path builds
    (``FrozenVlmo.attention_bias`` on CPU tensors: relative-position bias + -inf on padded text keys,
    multiway_transformer.py:88-118 / vlmo_module.py:807-814)."""
    import torch
    from vqattack_amd.attention import KeyHoleBias
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, vlmo_tiny
    model = FrozenVlmo(vlmo_tiny(), seed=0)
    masks = torch.tensor([[1, 1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1]])
    dense = model.attention_bias(masks)                     # host tensors: materialised per sample
    full = model.attention_bias(torch.ones_like(masks))     # no padding: the shared slab
    lengths = masks.sum(dim=1)
    hole = torch.stack([lengths, torch.full_like(lengths, masks.shape[1])], dim=1).to(torch.int32)
    for li in range(model.cfg.depth):
        assert not isinstance(dense[li], KeyHoleBias)
        got = KeyHoleBias(full[li], hole).dense()
        assert got.shape == dense[li].shape and torch.equal(got, dense[li])
