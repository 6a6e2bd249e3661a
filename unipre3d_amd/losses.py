"""Render losses that seed dL/dcolor.

Restates utils/loss_utils.py:17-45 (l1/l2/focal_l2) and the selection logic of
train_network.py:260-302 (`calculate_losses`) without the LPIPS term (its network weights are
not available offline; weight 0.01 and only after iteration 50 000 in the reference).
"""
from __future__ import annotations

import torch


def l1_loss(x, gt):
    return torch.abs(x - gt).mean()


def l2_loss(x, gt):
    return ((x - gt) ** 2).mean()


def focal_l2_weights(gt: torch.Tensor, bg_color, non_bg_rate: float, bg_rate: float) -> torch.Tensor:
    """(N,1,H,W) per-pixel weight: pixels whose 3 gt channels are all isclose(bg, atol=1e-6, rtol=1e-5)
    get 2*bg/(bg+non_bg), the rest 2*non_bg/(bg+non_bg) (utils/loss_utils.py:28-38)."""
    # 0-dim HOST tensors like the reference's `torch.tensor(bg_color[0])`: no H2D copy per call, so the loss can sit inside a HIP-graph capture
    # (a TENSOR background -- e.g. the batch's device-resident `bg` -- is indexed where it lives: float(c) on it would be one device->host
    # synchronisation per channel, which is exactly what a capture cannot contain)
    if torch.is_tensor(bg_color):
        bg = [bg_color[i].to(gt.dtype) for i in range(3)]
    else:
        bg = [torch.tensor(float(c), dtype=gt.dtype) for c in bg_color]
    is_bg = torch.isclose(gt[:, 0], bg[0], atol=1e-6) & torch.isclose(gt[:, 1], bg[1], atol=1e-6) & \
        torch.isclose(gt[:, 2], bg[2], atol=1e-6)
    w_non = 2 * non_bg_rate / (bg_rate + non_bg_rate)
    w_bg = 2 * bg_rate / (bg_rate + non_bg_rate)
    return torch.where(is_bg, w_bg, w_non).unsqueeze(1).to(gt.dtype)


def focal_l2_loss(x, gt, bg_color, non_bg_color_loss_rate, bg_color_loss_rate):
    w = focal_l2_weights(gt, bg_color, non_bg_color_loss_rate, bg_color_loss_rate)
    return (((x - gt) ** 2) * w).mean()


def render_loss(x, gt, kind: str = "focal_l2", white_background: bool = False, non_bg_color_loss_rate: float = 4,
                bg_color_loss_rate: float = 1):
    """train_network.py:260-302 minus LPIPS."""
    if kind == "focal_l2":
        bg = [1.0, 1.0, 1.0] if white_background else [0.0, 0.0, 0.0]
        return focal_l2_loss(x, gt, bg, non_bg_color_loss_rate, bg_color_loss_rate)
    if kind == "l1":
        return l1_loss(x, gt)
    return l2_loss(x, gt)
