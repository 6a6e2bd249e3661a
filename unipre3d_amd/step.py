"""One render-loss training step (SURVEY.md R1 -> R7/R2/R4 -> R8 -> R5): the hot loop of
Trainer.train_iteration (train_network.py:450-464) from the head output onward, with the per-object /
per-view Python loop replaced by one batched operator call."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import head as head_mod
from . import losses
from .synthetic import SyntheticBatch


def render_loss_forward(raw: torch.Tensor, batch: SyntheticBatch, H: int, W: int, input_images: int = 0,
                        loss_kind: str = "focal_l2", render_fn: Optional[Callable] = None, max_sh_degree: int = 1):
    """raw (B,23,P) head output -> (loss, rendered (B*V,3,H,W)).  `render_fn` defaults to the HIP batched
    renderer; tests may inject another differentiable renderer with the same signature."""
    if render_fn is None:
        from .renderer import render_views as render_fn
    if batch.level == "object":
        g = head_mod.process_object_output(raw, batch.center, batch.offset_scale, max_sh_degree)
    else:
        B, C, P = raw.shape
        flat = raw.permute(0, 2, 1).reshape(B * P, C)
        idx = torch.arange(B, device=raw.device).repeat_interleave(P)[:, None]
        lists = head_mod.process_scene_output(flat, batch.center.reshape(B * P, 3), idx, batch.offset_scale, max_sh_degree)
        g = {k: torch.stack(v) for k, v in lists.items()}
    rendered = render_fn(g, batch.world_view, batch.full_proj, batch.camera_center, batch.bg, batch.fov_deg, H, W,
                         input_images=input_images, max_sh_degree=max_sh_degree)
    gt = batch.gt[:, input_images:].reshape(-1, 3, H, W)
    white = bool(batch.bg[0].item() > 0.5) if loss_kind == "focal_l2" else False
    loss = losses.render_loss(rendered, gt, loss_kind, white_background=white)
    return loss, rendered


def train_step(model: torch.nn.Module, feats: torch.Tensor, batch: SyntheticBatch, optimizer: torch.optim.Optimizer, H: int,
               W: int, input_images: int = 0, loss_kind: str = "focal_l2", render_fn: Optional[Callable] = None,
               clip_grad: Optional[float] = 1.0, fused: bool = False, return_found_inf: bool = False):
    """zero_grad -> head -> render-loss -> backward (DDP all-reduce inside) -> clip -> optimizer step
    (train_network.py:329-352 without the per-parameter NaN scan's host syncs).
    fused=True: activations + batched render + loss run inside the HIP library (fused.render_loss_fused).
    return_found_inf=True: returns (loss, found_inf) where found_inf says whether the step was SKIPPED because a gradient held a
    NaN / Inf -- a bool on the host-read route, a 0-dim device tensor (1.0 = skipped; no host sync) on the deferred route -- so that
    callers can gate their scheduler / EMA update on it the way the reference's `continue` does (train_network.py:336-340)."""
    optimizer.zero_grad(set_to_none=True)
    if fused:
        from .fused import render_loss_fused
        head_out = model(feats, point_major=True)
        loss, _, _ = render_loss_fused(head_out, batch.center, batch.world_view, batch.full_proj, batch.camera_center, batch.gt,
                                       batch.bg, batch.fov_deg, H, W, level=batch.level, offset_scale=batch.offset_scale,
                                       loss_kind=loss_kind, input_images=input_images, return_images=False)
    else:
        raw = model(feats)
        loss, _ = render_loss_forward(raw, batch, H, W, input_images, loss_kind, render_fn)
    loss.backward()
    found_inf = False
    deferred = False
    if clip_grad is not None:
        from .gradcheck import check_and_clip_deferred, check_and_clip_gradients
        params = [p for p in model.parameters() if p.grad is not None]
        if params and params[0].grad.is_cuda and optimizer.defaults.get("fused"):
            # fused optimizer on a HIP device: the NaN / Inf decision stays on the device (`found_inf` makes the optimizer skip the
            # step exactly when the reference's `if not valid: skip` would, train_network.py:336-340) -- no host read in the step
            check_and_clip_deferred(params, optimizer, clip_grad)
            deferred = True
            if return_found_inf:
                found_inf = optimizer.found_inf.clone()         # (the state block is rewritten by the next step's stats pass)
        elif not check_and_clip_gradients(params, clip_grad):   # NaN/Inf: skip the step
            optimizer.zero_grad(set_to_none=True)
            return (loss.detach(), True) if return_found_inf else loss.detach()
    optimizer.step()
    if deferred:
        # the decision belongs to THIS step only: a later step that does not go through check_and_clip_deferred (clip_grad=None,
        # another code path) must not inherit a stale found_inf = 1 and silently skip every optimizer.step() after it
        optimizer.found_inf = None
        optimizer.grad_scale = None
    return (loss.detach(), found_inf) if return_found_inf else loss.detach()
