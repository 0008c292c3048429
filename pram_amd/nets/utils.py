"""Keypoint normalisation constants (behaviour of the reference's nets/utils.py:17-24)."""
from __future__ import annotations

import torch


def keypoint_norm_constants(image_shape):
    """(cx, cy, scale) such that normalize_keypoints(k) = (k - (cx, cy)) / scale.

    The reference unpacks ``_, _, height, width = image_shape`` (nets/utils.py:19) and the matcher call sites
    hand over (1, 3, width, height) (localization/singlemap3d.py:147): like the reference, the formula is
    applied to whatever tuple arrives — the swap is a pinned quirk, not something to fix."""
    height, width = float(image_shape[2]), float(image_shape[3])
    return width / 2.0, height / 2.0, max(width, height) * 0.7


def normalize_keypoints(kpts: torch.Tensor, image_shape) -> torch.Tensor:
    """Tensor form with the reference's arithmetic (subtract centre, divide by 0.7 * longer side)."""
    cx, cy, scale = keypoint_norm_constants(image_shape)
    return (kpts - kpts.new_tensor([cx, cy])) / kpts.new_tensor(scale)


def arange_like(x: torch.Tensor, dim: int) -> torch.Tensor:
    """0 .. x.shape[dim]-1 in x's dtype and device (nets/utils.py:13)."""
    return torch.arange(x.shape[dim], device=x.device, dtype=x.dtype)
