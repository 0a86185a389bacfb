"""Generate tests/golden/resize_golden.npz with Pillow itself (the third-party code the reference's transforms call).

    python -m tests.golden.make_resize_golden

Stores, per case, the seeded uint8 input image and Pillow's ``Image.resize((S, S), BICUBIC)`` output.
"""
import os

import numpy as np
import PIL
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_golden.npz")
# (name, H, W, S): down-scale, up-scale, mixed, one axis unchanged, odd sizes, the metric's 384 and the reference's 480
# (the 480 case up-scales a small image to keep the fixture small)
CASES = [("down_640x480_to_96", 48, 64, 10), ("vqa_480x640_to_384", 480, 640, 384), ("up_20x30_to_48", 20, 30, 48),
         ("mixed_100x40_to_64", 100, 40, 64), ("same_w_50x64_to_64", 50, 64, 64), ("odd_37x53_to_29", 37, 53, 29),
         ("ref_120x160_to_480", 120, 160, 480), ("tiny_3x5_to_8", 3, 5, 8), ("big_down_333x1000_to_32", 333, 1000, 32)]


def make_image(h, w, seed):
    r = np.random.RandomState(seed)
    base = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((np.sin(yy / 7.0) + np.cos(xx / 5.0)) * 60 + 128).clip(0, 255).astype(np.uint8)
    base[..., 1] = smooth                      # one smooth channel, two noisy ones (over/undershoot gets clipped)
    base[: max(h // 8, 1)] = 255               # saturated band: bicubic overshoot must clamp at 255
    base[-max(h // 8, 1):] = 0
    return base


def main():
    blob = {"pillow_version": np.array(PIL.__version__)}
    for i, (name, h, w, s) in enumerate(CASES):
        img = make_image(h, w, 100 + i)
        out = np.asarray(Image.fromarray(img, "RGB").resize((s, s), Image.BICUBIC))
        blob[name + "/in"], blob[name + "/out"] = img, out
        print(name, img.shape, "->", out.shape)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
