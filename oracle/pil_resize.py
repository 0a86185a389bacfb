"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the image input pipeline either side of the attack.

The reference turns a PIL RGB image into the white box's input with
``transforms.Resize((res, res), interpolation=Image.BICUBIC)`` -> ``ToTensor()`` -> ``Normalize(0.5, 0.5)``
(``ALBEF_attack/dataset/__init__.py:17,35-39``; ``vlmo/transforms/square_transform.py:11-18``).  The arithmetic lives in
third-party code that is not vendored in the reference tree:

  * ``torchvision.transforms.Resize`` on a PIL image = ``PIL.Image.resize(size, BICUBIC)`` (torchvision is absent here);
  * Pillow (pinned ``Pillow==8.3.1`` in ``VLMO_VQAttack/requirements.txt``), ``src/libImaging/Resample.c``:
    ``precompute_coeffs`` (double precision, bicubic a = -0.5, filter support stretched by the down-scale factor),
    ``normalize_coeffs_8bpc`` (fixed point, PRECISION_BITS = 32 - 8 - 2 = 22), horizontal pass then vertical pass, each
    rounding to uint8 through ``clip8``.

This file restates that published algorithm in numpy.  Pin: ``tests/golden/make_resize_golden.py`` runs Pillow itself
(12.2.0 in the build container; Resample.c's 8-bit path is unchanged since 8.3.1) on seeded images and stores its outputs
in ``tests/golden/resize_golden.npz``; ``tests/test_preprocess.py`` requires exact uint8 equality.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box (in0 = 0, in1 = in_size).
    Returns (bounds int32 (out, 2) = [xmin, count], kk int32 (out, ksize), ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img, out_h, out_w):
    """img uint8 (H, W, C) -> uint8 (out_h, out_w, C), like PIL.Image.resize((out_w, out_h), BICUBIC)."""
    h, w, c = img.shape
    cur = img
    if out_w != w:                                   # horizontal pass first (Resample.c ImagingResampleInner)
        bounds, kk, _ = precompute_coeffs(w, out_w)
        nxt = np.empty((h, out_w, c), dtype=np.uint8)
        for xo in range(out_w):
            x0, n = bounds[xo]
            acc = (cur[:, x0:x0 + n, :].astype(np.int64) * kk[xo, :n, None].astype(np.int64)).sum(axis=1)
            nxt[:, xo, :] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = nxt
    if out_h != h:
        bounds, kk, _ = precompute_coeffs(h, out_h)
        nxt = np.empty((out_h, cur.shape[1], c), dtype=np.uint8)
        for yo in range(out_h):
            y0, n = bounds[yo]
            acc = (cur[y0:y0 + n].astype(np.int64) * kk[yo, :n, None, None].astype(np.int64)).sum(axis=0)
            nxt[yo] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = nxt
    return cur


def to_tensor_normalize(img_u8, mean=0.5, std=0.5):
    """ToTensor + Normalize in fp32: (u8 / 255 - mean) / std, CHW."""
    t = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return (t - np.float32(mean)) / np.float32(std)


def preprocess(img_u8, size):
    return to_tensor_normalize(resize_bicubic_u8(img_u8, size, size))
