"""The sharded sweep driver end to end on the device (tiny encoders): bucketing, attack, scoring, .pt files, ASR."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("flavor", ["vlmo", "albef"])
def test_sweep_tiny(flavor, mixed, tmp_path):
    from vqattack_amd.attack.runner import AttackConfig
    from vqattack_amd.attack.sweep import run_sweep
    if flavor == "vlmo":
        from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters, vlmo_tiny
        cfg = vlmo_tiny()
        white = FrozenVlmo(cfg, seed=0).to(DEV)
        black = FrozenVlmo.finetuned_from(white, seed=1).to(DEV)
        adapters, text_len = VlmoAttackAdapters(white), cfg.max_text_len
    else:
        from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef, albef_tiny
        cfg = albef_tiny()
        white = FrozenAlbef(cfg, seed=0).to(DEV)
        black = FrozenAlbef.finetuned_from(white, seed=1).to(DEV)
        adapters, text_len = AlbefAttackAdapters(white), 8
    res = run_sweep(flavor, white, black, adapters, n_samples=11, batch=4, image_size=cfg.image_size,
                    text_len=text_len, device=DEV, config=AttackConfig(budget=8, sim_threshold=0.2),
                    save_dir=str(tmp_path), log_every=0, max_words=3, dual_every=3, mixed=mixed)
    assert res["n_total"] == 11 and res["n_local"] == 11
    assert 0.0 <= res["asr"] <= 1.0
    assert sorted(map(int, res["adv_text"])) == list(range(11))
    assert res["gradient_steps"] >= 11 * 8
    files = sorted(os.listdir(str(tmp_path)))
    assert len(files) == 11
    t = torch.load(os.path.join(str(tmp_path), "0.pt"))
    assert t.shape == (1, 3, cfg.image_size, cfg.image_size) and float(t.abs().max()) <= 1.0
