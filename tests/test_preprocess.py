"""Input pipeline: Pillow-exact bicubic resize + ToTensor + Normalize, and the .pt writer.

CPU: the numpy oracle against Pillow's own outputs (fixtures from tests/golden/make_resize_golden.py, exact uint8
equality) and the shipped host-side tap tables against the oracle's.  GPU: the HIP kernels against the oracle, bit-exact
(integer resampling; the fp32 normalisation is two correctly rounded ops).
"""
import os

import numpy as np
import pytest
import torch

from oracle import pil_resize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def resize_golden():
    with np.load(os.path.join(ROOT, "tests", "golden", "resize_golden.npz")) as z:
        return {k: z[k] for k in z.files}


def _cases(g):
    return sorted({k.split("/")[0] for k in g if "/" in k})


def test_oracle_matches_pillow_fixtures(resize_golden):
    names = _cases(resize_golden)
    assert len(names) >= 8
    for n in names:
        want = resize_golden[n + "/out"]
        got = pil_resize.resize_bicubic_u8(resize_golden[n + "/in"], want.shape[0], want.shape[1])
        assert np.array_equal(got, want), n


def test_oracle_matches_live_pillow_if_present():
    Image = pytest.importorskip("PIL.Image")
    r = np.random.RandomState(0)
    for (h, w, s) in [(17, 23, 31), (64, 48, 20), (9, 200, 33)]:
        img = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img, "RGB").resize((s, s), Image.BICUBIC))
        assert np.array_equal(pil_resize.resize_bicubic_u8(img, s, s), want)


@pytest.mark.parametrize("sizes", [(640, 384), (480, 384), (20, 48), (333, 32), (64, 64), (5, 8), (1000, 480)])
def test_shipped_tap_tables_equal_oracle(sizes):
    from vqattack_amd.preprocess import resample_tables
    b0, k0, n0 = pil_resize.precompute_coeffs(*sizes)
    b1, k1, n1 = resample_tables(*sizes)
    assert n0 == n1 and np.array_equal(b0, b1) and np.array_equal(k0, k1)
    assert (k1.sum(axis=1) - (1 << 22)).__abs__().max() <= k1.shape[1]      # taps sum to ~1.0 in fixed point


@pytest.mark.gpu
def test_hip_preprocess_bitexact(resize_golden):
    from vqattack_amd.preprocess import ImagePreprocessor
    names = _cases(resize_golden)
    by_size = {}
    for n in names:
        by_size.setdefault(resize_golden[n + "/out"].shape[0], []).append(n)
    for size, group in by_size.items():
        pre = ImagePreprocessor(size, "cuda:0")
        imgs = [resize_golden[n + "/in"] for n in group]
        out = pre(imgs)                                           # host uint8 arrays, mixed sizes, one batch
        assert out.shape == (len(group), 3, size, size)
        for i, n in enumerate(group):
            want = pil_resize.to_tensor_normalize(resize_golden[n + "/out"])
            assert np.array_equal(out[i].cpu().numpy().view(np.uint32), want.view(np.uint32)), n
        # device-resident uint8 input and a caller-provided output slot
        dev_in = [torch.from_numpy(resize_golden[n + "/in"]).to("cuda:0") for n in group]
        slot = torch.zeros(len(group), 3, size, size, device="cuda:0")
        assert torch.equal(pre(dev_in, out=slot), out)


@pytest.mark.gpu
def test_hip_preprocess_feeds_attack_range_check():
    """Normalised pixels lie in [-1, 1]: exactly the clip range the attack's sanity flag checks."""
    from vqattack_amd import ops
    from vqattack_amd.preprocess import ImagePreprocessor
    r = np.random.RandomState(3)
    imgs = [r.randint(0, 256, (50 + 7 * i, 40 + 11 * i, 3)).astype(np.uint8) for i in range(4)]
    imgs[0][:] = 255
    imgs[1][:] = 0
    x = ImagePreprocessor(32, "cuda:0")(imgs)
    assert float(x.max()) == 1.0 and float(x.min()) == -1.0
    flag = ops.new_flag("cuda:0")
    ops.linf_init(x, None, 0.125, -1, 1, flag=flag)
    assert int(flag.item()) == 0


@pytest.mark.gpu
def test_adv_image_writer_roundtrip(tmp_path):
    from vqattack_amd.preprocess import AdvImageWriter
    adv = torch.randn(5, 3, 16, 16, device="cuda:0")
    w = AdvImageWriter(str(tmp_path), "cuda:0")
    w.write(adv[:3], [11, 12, 13])
    w.write(adv[3:], [14, 15])
    w.close()
    for i, q in enumerate([11, 12, 13, 14, 15]):
        t = torch.load(os.path.join(str(tmp_path), "{}.pt".format(q)))
        assert t.shape == (1, 3, 16, 16) and t.dtype == torch.float32 and not t.is_cuda
        assert torch.equal(t[0], adv[i].cpu())
