"""Keypoint normalisation constants (reference: nets/utils.py:13-24)."""
from __future__ import annotations

import torch


def arange_like(x, dim: int):
    return x.new_ones(x.shape[dim]).cumsum(0) - 1


def keypoint_norm_constants(image_shape):
    """(cx, cy, scale) such that normalize_keypoints(k) = (k - (cx, cy)) / scale.

    nets/utils.py:19 unpacks ``_, _, height, width = image_shape``; the matcher call sites hand
    over (1, 3, width, height) (localization/singlemap3d.py:147) — like the reference, the
    formula is applied to whatever tuple arrives."""
    _, _, height, width = image_shape
    w, h = float(width), float(height)
    return w / 2.0, h / 2.0, max(w, h) * 0.7


def normalize_keypoints(kpts: torch.Tensor, image_shape) -> torch.Tensor:
    """Tensor form, same arithmetic as the reference (used by callers that want the values)."""
    cx, cy, scale = keypoint_norm_constants(image_shape)
    center = kpts.new_tensor([cx, cy])
    return (kpts - center) / kpts.new_tensor(scale)
