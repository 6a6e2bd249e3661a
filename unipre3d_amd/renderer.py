"""Render boundary: counterparts of the reference's Python wrappers.

  render_predicted  -- same name, arguments and result dict as gaussian_renderer/__init__.py:13-104
                       (one view; calls the per-view operator exactly like the reference does).
  render_views      -- what Trainer.render_validation_views (train_network.py:392-448) computes:
                       rendered (B*V',3,H,W) in item-major / view-minor order for the supervised views
                       input_images..n_views-1, but with ONE batched operator call instead of B*V'
                       Python-level launches, no per-view host->device camera copies and no torch.stack.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import head
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_batched


def _resolution(cfg):
    if hasattr(cfg.data, "training_resolution"):
        r = int(cfg.data.training_resolution)
        return r, r
    return int(cfg.data.training_height), int(cfg.data.training_width)


def render_predicted(pc: Dict[str, torch.Tensor], world_view_transform, full_proj_transform, camera_center,
                     bg_color: torch.Tensor, cfg, scaling_modifier=1.0, override_color=None, focals_pixels=None):
    """Drop-in for gaussian_renderer/__init__.py:13-104."""
    screenspace_points = torch.zeros_like(pc["xyz"], dtype=pc["xyz"].dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    if focals_pixels is None:
        tanfovx = math.tan(cfg.data.fov * math.pi / 360)
        tanfovy = math.tan(cfg.data.fov * math.pi / 360)
    else:
        res = cfg.data.training_resolution
        # tan(focal2fov(f, res)) with focal2fov = 2 atan(res / 2f)  (utils/graphics_utils.py:89-90)
        tanfovx = math.tan(2 * math.atan(res / (2 * focals_pixels[0].item())))
        tanfovy = math.tan(2 * math.atan(res / (2 * focals_pixels[1].item())))
    H, W = _resolution(cfg)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=world_view_transform, projmatrix=full_proj_transform, sh_degree=cfg.model.max_sh_degree,
        campos=camera_center, prefiltered=False, debug=False, antialiasing=True)
    rasterizer = GaussianRasterizer(raster_settings=settings)
    shs = colors_precomp = None
    if override_color is None:
        shs = head.concat_sh(pc["features_dc"], pc.get("features_rest"))
    else:
        colors_precomp = override_color
    rendered_image, radii, _ = rasterizer(means3D=pc["xyz"], means2D=screenspace_points, shs=shs,
                                          colors_precomp=colors_precomp, opacities=pc["opacity"], scales=pc["scaling"],
                                          rotations=pc["rotation"], cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


def render_views(gaussians: Dict[str, torch.Tensor], world_view: torch.Tensor, full_proj: torch.Tensor,
                 camera_center: torch.Tensor, bg: torch.Tensor, fov_deg: float, H: int, W: int, input_images: int = 0,
                 max_sh_degree: int = 1, scaling_modifier: float = 1.0) -> torch.Tensor:
    """gaussians: dict of (B,P,...) tensors (head.process_object_output) -- or, scene level, of per-item LISTS of (M_i,...)
    tensors as GaussianSplatPredictor returns them there (model/gaussian_predictor.py:331-364; head.process_scene_output): the
    reference's own loop cannot take those (`v.shape` on a list, train_network.py:423), eval.py:97-99 indexes them per item;
    here the sets of different sizes are packed and rendered in ONE launch sequence.  Cameras (B,Vtot,...).
    Returns (B*(Vtot-input_images), 3, H, W), ordered like train_network.py:418-446."""
    wv, fp, cc = world_view[:, input_images:], full_proj[:, input_images:], camera_center[:, input_images:]
    t = math.tan(fov_deg * math.pi / 360)  # same tan for x and y (gaussian_renderer/__init__.py:35-37)
    if isinstance(gaussians["xyz"], (list, tuple)):
        sizes = [int(x.shape[0]) for x in gaussians["xyz"]]
        B = len(sizes)
        cat = lambda k: torch.cat(list(gaussians[k]), dim=0)
        rest = gaussians.get("features_rest")
        shs = torch.cat([head.concat_sh(dc, rest[i] if rest is not None else None) for i, dc in enumerate(gaussians["features_dc"])], dim=0)
        color, _, _ = rasterize_gaussians_batched(
            cat("xyz"), cat("opacity"), wv, fp, cc, bg, H, W, t, t, shs=shs, scales=cat("scaling"), rotations=cat("rotation"),
            sh_degree=max_sh_degree, scale_modifier=scaling_modifier, antialiasing=True, sizes=sizes)
        return color.reshape(B * wv.shape[1], 3, H, W)
    B = gaussians["xyz"].shape[0]
    shs = head.concat_sh(gaussians["features_dc"], gaussians.get("features_rest"))
    color, _, _ = rasterize_gaussians_batched(
        gaussians["xyz"], gaussians["opacity"], wv, fp, cc, bg, H, W, t, t, shs=shs, scales=gaussians["scaling"],
        rotations=gaussians["rotation"], sh_degree=max_sh_degree, scale_modifier=scaling_modifier, antialiasing=True)
    return color.reshape(B * wv.shape[1], 3, H, W)
