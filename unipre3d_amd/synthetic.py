"""Seeded synthetic inputs of the reference's shapes -- there are no datasets
on the GPU box.  Reference-faithful regime: N(0,1) head output pushed through the reference's
activations (head.py, R1), cameras from the reference's matrix formulas (cameras.py, R0).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from . import cameras, head

# BASELINE.json configs: name -> sizes per GPU
CONFIGS = {
    "C1": dict(level="object", P=128, H=128, W=128, B=2, V=4),
    "C2": dict(level="object", P=128, H=256, W=256, B=32, V=4),
    "C3": dict(level="object", P=2048, H=256, W=256, B=16, V=4),
    "C4": dict(level="scene", P=40000, H=480, W=640, B=2, V=8),
    "C5": dict(level="scene", P=200000, H=480, W=640, B=1, V=8),
    # SURVEY's C4 / C5 rows read "~40 k voxels + fused pixel-Gaussians" (fusion/point_fusion.py:159-168 concatenates the voxels of the
    # unprojected pixels of the reference views, <= 8 x 480 x 640 before the 2 cm grid sampling): the same shapes with that share added
    "C4_fused": dict(level="scene", P=120000, H=480, W=640, B=2, V=8),
    "C5_fused": dict(level="scene", P=350000, H=480, W=640, B=1, V=8),
}


@dataclass
class SyntheticBatch:
    raw: torch.Tensor            # (B, 23, P) head output  (leaf for autograd)
    center: torch.Tensor         # (B, P, 3)
    world_view: torch.Tensor     # (B, V, 4, 4)
    full_proj: torch.Tensor      # (B, V, 4, 4)
    camera_center: torch.Tensor  # (B, V, 3)
    gt: torch.Tensor             # (B, V, 3, H, W)
    bg: torch.Tensor             # (3,)
    fov_deg: float
    level: str
    offset_scale: float

    def to(self, device):
        return SyntheticBatch(**{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def make_batch(B: int, P: int, V: int, H: int, W: int, level: str = "object", seed: int = 42,
               compact: bool = False, bg_fraction: float = 0.3) -> SyntheticBatch:
    """compact=True is the secondary 'compact-splat' regime: scale = exp(N(-4, 0.5))."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(B, 23, P, generator=g)
    if compact:
        raw[:, 4:7] = -4.0 + 0.5 * raw[:, 4:7]
    if level == "object":
        d = torch.randn(B, P, 3, generator=g)
        r = torch.rand(B, P, 1, generator=g) ** (1 / 3) * 0.5
        center = d / d.norm(dim=-1, keepdim=True) * r
        cams = [cameras.orbit_cameras(V, cameras.OBJECT_CAMERA_DISTANCE, cameras.OBJECT_FOV_DEG,
                                      cameras.OBJECT_ZNEAR, cameras.OBJECT_ZFAR, g) for _ in range(B)]
        fov, bgv, off = cameras.OBJECT_FOV_DEG, 0.0, 1.0
    else:
        box = torch.tensor([6.0, 5.0, 3.0])
        center = torch.rand(B, P, 3, generator=g) * box
        cams = [cameras.room_cameras(V, generator=g) for _ in range(B)]
        fov, bgv, off = cameras.SCENE_FOV_DEG, 1.0, 0.2
    wv = torch.stack([c[0] for c in cams]); fp = torch.stack([c[1] for c in cams]); cc = torch.stack([c[2] for c in cams])
    gt = torch.rand(B, V, 3, H, W, generator=g)
    is_bg = torch.rand(B, V, 1, H, W, generator=g) < bg_fraction
    gt = torch.where(is_bg, torch.full_like(gt, bgv), gt)
    return SyntheticBatch(raw=raw, center=center, world_view=wv, full_proj=fp, camera_center=cc, gt=gt,
                          bg=torch.full((3,), bgv), fov_deg=fov, level=level, offset_scale=off)


def gaussians_from_batch(batch: SyntheticBatch, max_sh_degree: int = 1) -> Dict[str, torch.Tensor]:
    """Head activations (R1).  Object level uses the reference's object branch; scene level applies the
    scene branch's per-quaternion normalisation on the same (B,23,P) layout."""
    if batch.level == "object":
        return head.process_object_output(batch.raw, batch.center, batch.offset_scale, max_sh_degree)
    B, C, P = batch.raw.shape
    flat = batch.raw.permute(0, 2, 1).reshape(B * P, C)
    idx = torch.arange(B, device=flat.device).repeat_interleave(P)[:, None]
    lists = head.process_scene_output(flat, batch.center.reshape(B * P, 3), idx, batch.offset_scale, max_sh_degree)
    return {k: torch.stack(v) for k, v in lists.items()}


def single_view_scene(P: int = 64, H: int = 48, W: int = 64, seed: int = 0, level: str = "object",
                      compact: bool = False, sh_degree: int = 1, unit_quats: bool = False):
    """Small one-view scene for unit tests: dict of float32 tensors in the operator's own layout."""
    b = make_batch(1, P, 1, H, W, level=level, seed=seed, compact=compact)
    g = gaussians_from_batch(b, 1)
    gen = torch.Generator().manual_seed(seed + 1000)
    M = (sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=gen) * (0.6 if sh_degree else 1.0)
    shs[:, 0] = g["features_dc"][0, :, 0]
    rot = g["rotation"][0]
    if unit_quats:
        rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=gen), dim=-1)
    import math
    t = math.tan(b.fov_deg * math.pi / 360)
    return dict(means3D=g["xyz"][0].contiguous(), opacities=g["opacity"][0].contiguous(),
                scales=g["scaling"][0].contiguous(), rotations=rot.contiguous(), shs=shs.contiguous(),
                viewmatrix=b.world_view[0, 0].contiguous(), projmatrix=b.full_proj[0, 0].contiguous(),
                campos=b.camera_center[0, 0].contiguous(), bg=b.bg.clone(), image_height=H, image_width=W,
                tanfovx=t, tanfovy=t, sh_degree=sh_degree)
