"""ALBEF's per-forward random token masking at batch > 1 (``ALBEF_attack/models/model_pretrain.py:130-132``, ``mask``
:309-332).  At batch 1 the product and the oracle issue the same sequence of draws and are compared sample-wise
(tests/test_attack_batched_parity.py); for B > 1 the product draws ONE (B, L) mask where the reference's batch-1 loop would
draw B separate ones, so the runs are statistically -- not sample-wise -- equivalent.  This is the statistical statement,
tested: every sample is masked at the reference's rate, independently of its batch neighbours; [CLS] and padding are never
touched; 80 % of the selected tokens become [MASK], 10 % a random word, 10 % stay.  Both draw paths: the seeded host
generator (parity runs) and the device draw (production runs, no host round trip per forward).
"""
import numpy as np
import pytest
import torch


def _model():
    from vqattack_amd.whitebox.albef import FrozenAlbef, albef_tiny
    return FrozenAlbef(albef_tiny(mlm_probability=0.15), seed=0)


def _ids(batch, length, seed=0):
    r = np.random.RandomState(seed)
    ids = np.zeros((batch, length), dtype=np.int64)
    for s in range(batch):
        n = int(r.randint(4, length - 1))
        ids[s, 0] = 101
        ids[s, 1:1 + n] = r.randint(1000, 30522, n)
        ids[s, 1 + n] = 102
    return torch.from_numpy(ids)


def _check_statistics(draw, ids, n_draws):
    """``draw() -> masked ids``; returns after asserting the per-sample rates."""
    cfg_mask, b = 103, ids.shape[0]
    eligible = (ids != 0) & (ids != 101)
    changed = torch.zeros(ids.shape, dtype=torch.float64)
    to_mask = torch.zeros(ids.shape, dtype=torch.float64)
    pair = torch.zeros(b, b, dtype=torch.float64)            # co-occurrence of "sample i has a masked token" events
    for _ in range(n_draws):
        out = draw().cpu()
        diff = out != ids
        assert not bool((diff & ~eligible).any()), "[CLS] or padding was modified"
        changed += diff
        to_mask += (out == cfg_mask) & diff
        any_row = diff.any(dim=1).double()
        pair += any_row[:, None] * any_row[None, :]
    n_elig = eligible.sum(dim=1).double()
    # a selected token changes with probability 0.8 + 0.1 * (1 - 1/V) ~ 0.9: changed rate = 0.15 * 0.9 = 0.135 per token
    rate = changed.sum(dim=1) / (n_elig * n_draws)
    sigma = np.sqrt(0.135 * 0.865 / (float(n_elig.min()) * n_draws))
    assert float((rate - 0.135).abs().max()) <= 5 * sigma, (rate.tolist(), sigma)       # EVERY sample, not the batch mean
    share_mask = float(to_mask.sum() / changed.sum())
    assert abs(share_mask - 0.8 / 0.9) <= 0.02, share_mask                             # 80 % [MASK] of the 90 % that change
    # independence across samples: P(i and j both touched) = P(i) P(j)
    p = pair.diagonal() / n_draws
    off = pair / n_draws - p[:, None] * p[None, :]
    off.fill_diagonal_(0)
    assert float(off.abs().max()) <= 5 * 0.5 / np.sqrt(n_draws)


def test_host_draw_masks_every_sample_at_the_reference_rate():
    model = _model()
    ids = _ids(16, 14)
    model.seed_masking(1234)
    _check_statistics(lambda: model.mask_tokens(ids), ids, 600)


@pytest.mark.gpu
def test_device_draw_masks_every_sample_at_the_reference_rate():
    model = _model().to("cuda:0")
    ids = _ids(16, 14).to("cuda:0")
    model.seed_masking(None)                                 # production path: the draw happens on the device
    torch.manual_seed(7)
    out = model.mask_tokens(ids)
    assert out.device == ids.device and out.data_ptr() != ids.data_ptr()
    _check_statistics(lambda: model.mask_tokens(ids), ids.cpu(), 600)
