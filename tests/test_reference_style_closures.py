"""CPU check of ``vqattack_amd/whitebox/reference_style.py``: the reference-form closures (``self.batch``, batch 1, packed
plain tensors) return what the oracle's restatement of the same reference members returns (``oracle/adapters_ref.py``,
itself pinned by the reference's methods executed from source: ``pack_vlmo_*`` of tests/golden/text_golden.npz)."""
import torch


def _text(n_real, length):
    ids = torch.zeros(1, length, dtype=torch.long)
    ids[0, :n_real] = torch.tensor([101] + list(range(1000, 1000 + n_real - 2)) + [102])
    return ids, (ids != 0).long()


def test_vlmo_reference_closures_equal_the_oracle_adapters():
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.whitebox.reference_style import VlmoReferenceClosures
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, vlmo_tiny
    cfg = vlmo_tiny()
    model = FrozenVlmo(cfg, seed=0)
    ids, masks = _text(5, cfg.max_text_len)
    ids_mlm = ids.clone()
    ids_mlm[0, 2] = 103
    g = torch.Generator().manual_seed(0)
    img = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    batch = dict(text_ids=ids, text_masks=masks, text_ids_mlm=ids_mlm, text_mask_mlm=masks)
    me, ref = VlmoReferenceClosures(model, batch), VlmoRefAdapters(model, ids, masks, ids_mlm, masks)
    emb = model.text_embeddings(ids)
    # round 6: with a prefix mask the closures do not run the all-padding columns through the encoder (masked keys, dropped
    # rows): the same values to fp32 rounding of a shorter softmax, not the same bits; the MLM closure's logits at the
    # dropped (padded) positions are zeros where the reference scores padding -- those positions carry ignore_index labels
    def same(a, b):
        return a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    for k, (got, want) in enumerate(((me.pgd_attack(img), ref.pgd_attack(img)),
                                     (me.pgd_attack_vl([img, emb]), ref.pgd_attack_vl([img, emb])),
                                     (me.pgd_mlm_attack(img), ref.pgd_mlm_attack(img)),
                                     (me.Gen_ori_feats(img), ref.gen_ori_feats(img)))):
        assert isinstance(got, list) and len(got) == 3
        if k == 2:
            assert got[0].shape == want[0].shape and same(got[0][:, :5], want[0][:, :5]) and not bool(got[0][:, 5:].any())
            got, want = got[1:], want[1:]
        assert all(same(a, b) for a, b in zip(got, want))
    # a mask that is not a prefix keeps the reference's gather + cat form (all 40 columns encoded)
    holes = masks.clone()
    holes[0, 2] = 0
    me2, ref2 = VlmoReferenceClosures(model, dict(text_ids=ids, text_masks=holes)), VlmoRefAdapters(model, ids, holes)
    assert all(same(a, b) for a, b in zip(me2.pgd_attack(img), ref2.pgd_attack(img)))
    assert got[2].shape[1] == 5 + cfg.n_image_tokens                 # padded text tokens dropped, image part after the text
    # the closure reads self.batch at call time: the orchestrator swaps the text between operator calls
    ids2, masks2 = _text(7, cfg.max_text_len)
    batch.update(text_ids=ids2, text_masks=masks2)
    assert me.pgd_attack(img)[2].shape[1] == 7 + cfg.n_image_tokens


def test_albef_reference_closures_equal_the_oracle_adapters():
    from oracle.adapters_ref import AlbefRefAdapters
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_tiny
    from vqattack_amd.whitebox.reference_style import AlbefReferenceClosures
    cfg = albef_tiny()
    model = FrozenAlbef(cfg, seed=0)
    ids, masks = _text(6, 8)
    g = torch.Generator().manual_seed(1)
    img = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    me = AlbefReferenceClosures(model, dict(text_ids=ids, text_masks=masks, text_ids_mlm=ids, text_mask_mlm=masks))
    ref = AlbefRefAdapters(model, ids, masks)
    emb = model.text_embeddings(ids)
    for seed, (f, h) in enumerate(((me.pgd_attack, ref.pgd_attack), (me.pgd_mlm_attack, ref.pgd_mlm_attack))):
        model.seed_masking(seed)                                       # the per-forward random token masking, seeded alike
        got = f(img)
        model.seed_masking(seed)
        want = h(img)
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    model.seed_masking(3)
    got = me.pgd_attack_vl([img, emb])
    model.seed_masking(3)
    want = ref.pgd_attack_vl([img, emb])
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    model.seed_masking(4)
    img_feats, txt_feats = me.Gen_ori_feats(img)                       # the reference returns (image, text): adv_attack.py:118
    model.seed_masking(4)
    want = ref.gen_ori_feats(img)
    assert torch.equal(txt_feats, want[0]) and torch.equal(img_feats, want[1])
    assert img_feats.shape[0] == cfg.vit_depth + 1
