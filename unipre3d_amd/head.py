"""Gaussian head post-processing: network output -> rasterizer inputs.

Restates model/gaussian_predictor.py:174-181 (channel split [3,1,3,4,3,9]), :249-254
(activations) and :279-328, :366-420 (object/scene output assembly), including the
parity-relevant quirk that at object level `F.normalize(rotation, dim=-1)` is applied to a
(B,4,N) tensor, i.e. normalises ACROSS POINTS, not per quaternion (:254, :318).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SPLIT_DIMS = (3, 1, 3, 4, 3, 9)  # xyz, opacity, scaling, rotation, features_dc, features_rest (sh degree 1)


def split_dims(max_sh_degree: int = 1):
    rest = ((max_sh_degree + 1) ** 2 - 1) * 3
    return (3, 1, 3, 4, 3) + ((rest,) if rest > 0 else ())


def _flatten_vector(x: torch.Tensor) -> torch.Tensor:
    return x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)


def scaling_activation(x):
    return torch.exp(torch.clamp(x, -1, 20))


def process_object_output(raw: torch.Tensor, center: torch.Tensor, offset_scale: float = 1.0,
                          max_sh_degree: int = 1, isotropic: bool = False) -> Dict[str, torch.Tensor]:
    """raw (B, C, N) head output, center (B, N, >=3) -> dict of (B, N, ...) Gaussian parameters
    (model/gaussian_predictor.py:279-328, object branch)."""
    parts = raw.split(list(split_dims(max_sh_degree)), dim=1)
    xyz_raw, opacity, scaling, rotation, features_dc = parts[:5]
    pos = torch.tanh(xyz_raw) * offset_scale
    pos = pos.permute(0, 2, 1) + center[:, :, :3]
    if isotropic:
        scaling = scaling[:, :1].expand(-1, 3, -1)
    out = {
        "xyz": pos,
        "opacity": _flatten_vector(torch.sigmoid(opacity)),
        "scaling": _flatten_vector(scaling_activation(scaling)),
        # NB: normalises over the LAST dim of (B,4,N) = across points (reference quirk, R1)
        "rotation": _flatten_vector(F.normalize(rotation, dim=-1, eps=1e-6)),
        "features_dc": _flatten_vector(features_dc).unsqueeze(2),
    }
    if max_sh_degree > 0:
        fr = _flatten_vector(parts[5])
        out["features_rest"] = fr.reshape(*fr.shape[:2], -1, 3)
    else:
        out["features_rest"] = torch.zeros(out["features_dc"].shape[0], 0, 3, dtype=raw.dtype, device=raw.device)
    return {k: v.contiguous() for k, v in out.items()}


def process_scene_output(raw: torch.Tensor, center: torch.Tensor, indices: torch.Tensor, offset_scale: float = 0.2,
                         max_sh_degree: int = 1) -> Dict[str, List[torch.Tensor]]:
    """raw (M, C) per-voxel head output, center (M, >=3), indices (M, >=1) with batch id in column 0
    -> dict of per-scene lists (model/gaussian_predictor.py:331-364, scene branch; per-quaternion
    normalisation is correct there because tensors are (M_i, 4))."""
    parts = raw.split(list(split_dims(max_sh_degree)), dim=1)
    xyz_raw, opacity, scaling, rotation, features_dc = parts[:5]
    pos = torch.tanh(xyz_raw) * offset_scale + center[:, :3]
    n = int(indices[:, 0].max().item()) + 1
    keys = ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest")
    out = {k: [None] * n for k in keys}
    for b in range(n):
        m = indices[:, 0] == b
        out["xyz"][b] = pos[m]
        out["opacity"][b] = torch.sigmoid(opacity[m])
        out["scaling"][b] = scaling_activation(scaling[m])
        out["rotation"][b] = F.normalize(rotation[m], dim=-1, eps=1e-6)
        out["features_dc"][b] = features_dc[m].unsqueeze(1)
        if max_sh_degree > 0:
            fr = parts[5][m]
            out["features_rest"][b] = fr.reshape(fr.shape[0], -1, 3)
        else:
            out["features_rest"][b] = torch.zeros(int(m.sum()), 0, 3, dtype=raw.dtype, device=raw.device)
    return out


def concat_sh(features_dc: torch.Tensor, features_rest: Optional[torch.Tensor]) -> torch.Tensor:
    """(..., 1, 3) + (..., K-1, 3) -> (..., K, 3) contiguous (gaussian_renderer/__init__.py:80-84)."""
    if features_rest is None or features_rest.shape[-2] == 0:
        return features_dc.contiguous()
    return torch.cat([features_dc, features_rest], dim=-2).contiguous()
