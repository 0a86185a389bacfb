"""Patch-major image layout for the PGD state.

ViT-style white boxes start with ``x.reshape(B,3,g,p,g,p).permute(0,2,4,1,3,5).reshape(B, g*g, 3*p*p) @ W`` -- a full
copy of the image in the forward and of its gradient in the backward, every PGD iteration (16 B/element, as much as
the fused step kernel itself moves).  Every image kernel of this library is elementwise or a per-sample reduction, i.e.
indifferent to the order of the elements inside a sample, so an attack can keep ``x``, ``ori_x`` and the perturbation in
patch-major layout ``(B, g*g, 3*p*p)`` for its whole duration: the encoder consumes it without the permute, autograd
returns the gradient in the same layout, and the image is converted back once at the end.
"""
import torch


def to_patches(images, patch):
    """(B, 3, H, W) -> (B, (H/p)*(W/p), 3*p*p), the row layout of the patch-embedding GEMM."""
    b, c, h, w = images.shape
    gh, gw = h // patch, w // patch
    return images.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * patch * patch) \
        .contiguous()


def from_patches(patches, patch, height, width, channels=3):
    """Inverse of ``to_patches``."""
    b = patches.shape[0]
    gh, gw = height // patch, width // patch
    return patches.reshape(b, gh, gw, channels, patch, patch).permute(0, 3, 1, 4, 2, 5).reshape(b, channels, height, width) \
        .contiguous()
